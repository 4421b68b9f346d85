#!/bin/sh
# GPU box: per-kernel times of the bench MSM (rocprofv3 --kernel-trace --stats) -> gpurun_out/kstats/<tag>_stats.csv ; usage: tools/kstats.sh <tag> [bench args]
cd "$(dirname "$0")/.."
R=$PWD; TAG=${1:-run}; shift
O=$R/gpurun_out/kstats; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/$TAG -o $TAG -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-prove --no-extras --no-traffic "$@" > $O/$TAG.json 2>$O/$TAG.err
cd $R
python3 - "$O" "$TAG" <<'PY'
import sqlite3, glob, os, csv, collections, sys
O, tag = sys.argv[1], sys.argv[2]
dbs = glob.glob(f"{O}/{tag}/**/*_results.db", recursive=True)
con = sqlite3.connect(dbs[0]); cur = con.cursor()
t = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [x for x in t if "kernel_dispatch" in x][0]; ks = [x for x in t if "kernel_symbol" in x][0]
agg = collections.defaultdict(list)
for n, dt in cur.execute(f"select s.display_name, d.end - d.start from {kd} d join {ks} s on d.kernel_id = s.id"): agg[n.replace("(anonymous namespace)::", "").split("(")[0].replace("void mnt753::", "").replace("void ", "")].append(dt)
total = sum(sum(v) for v in agg.values()) or 1
with open(f"{O}/{tag}_stats.csv", "w", newline="") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([n, len(v), sum(v), sum(v) / len(v), round(100 * sum(v) / total, 2), min(v), max(v)])
        if sum(v) / total > 0.004: print(f"{n[:64]:64s} calls {len(v):4d} avg_us {sum(v)/len(v)/1e3:10.1f} min_us {min(v)/1e3:10.1f} max_us {max(v)/1e3:10.1f} total_ms {sum(v)/1e6:9.2f}")
con.close()
PY
rm -rf $O/$TAG
