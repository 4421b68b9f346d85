#!/bin/sh
# GPU box: per-kernel times of one MSM configuration of tools/slice_sweep.py under rocprofv3 --kernel-trace
#   MNT753_MSM_PRE_C=16 sh tools/experiments/kstats_sweep.sh <tag> <curve:group:size>
cd "$(dirname "$0")/../.."
R=$PWD; TAG=$1; CFG=$2
O=$R/gpurun_out/kstats; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/$TAG -o $TAG -- python3 $R/tools/slice_sweep.py --quick --configs $CFG --out $O/$TAG.json > $O/$TAG.log 2>&1
cd $R
python3 - "$O" "$TAG" <<'PY'
import sqlite3, glob, csv, collections, sys
O, tag = sys.argv[1], sys.argv[2]
dbs = glob.glob(f"{O}/{tag}/**/*_results.db", recursive=True)
con = sqlite3.connect(dbs[0]); cur = con.cursor()
t = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [x for x in t if "kernel_dispatch" in x][0]; ks = [x for x in t if "kernel_symbol" in x][0]
agg = collections.defaultdict(list)
for n, dt in cur.execute(f"select s.display_name, d.end - d.start from {kd} d join {ks} s on d.kernel_id = s.id"): agg[n.replace("(anonymous namespace)::", "").split("(")[0].replace("void mnt753::", "").replace("void ", "")].append(dt)
print("==", tag)
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if "precompute" in n or "synth" in n: continue
    print(f"{n[:64]:64s} calls {len(v):4d} avg_us {sum(v)/len(v)/1e3:10.1f} min_us {min(v)/1e3:10.1f} max_us {max(v)/1e3:10.1f} total_ms {sum(v)/1e6:9.2f}")
con.close()
PY
rm -rf $O/$TAG
