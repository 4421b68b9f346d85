#!/bin/sh
# GPU box: one rocprofv3 --pmc pass (with --kernel-trace for the durations) over a short bench run.
#   tools/pmc.sh <tag> <counter> [<counter> ...]     -> gpurun_out/pmc/<tag>.csv (kernel, avg duration, avg counter values)
cd "$(dirname "$0")/.."
R=$PWD; TAG=$1; shift
O=$R/gpurun_out/pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" -d $O/$TAG -o $TAG -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prove --no-extras --no-traffic > $O/$TAG.log 2>&1
cd $R
python3 - "$O" "$TAG" <<'PY'
import sqlite3, glob, os, csv, collections, sys
O, tag = sys.argv[1], sys.argv[2]
dbs = glob.glob(f"{O}/{tag}/**/*_results.db", recursive=True)
con = sqlite3.connect(dbs[0]); cur = con.cursor()
t = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [x for x in t if "kernel_dispatch" in x][0]; ks = [x for x in t if "kernel_symbol" in x][0]
pm = [x for x in t if "pmc_event" in x][0]; pi = [x for x in t if "info_pmc" in x][0]
dur = collections.defaultdict(list)
for n, dt in cur.execute(f"select s.display_name, d.end - d.start from {kd} d join {ks} s on d.kernel_id = s.id"): dur[n.split("(")[0].replace("void mnt753::", "")].append(dt)
# counter values: summed over the instances (dimensions) of one dispatch, then averaged over dispatches
per = collections.defaultdict(lambda: collections.defaultdict(float))
for n, sym, val, ev in cur.execute(f"select s.display_name, p.symbol, e.value, d.id from {pm} e join {pi} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id"):
    per[(n.split("(")[0].replace("void mnt753::", ""), sym)][ev] += val
syms = sorted({k[1] for k in per})
with open(f"{O}/{tag}.csv", "w", newline="") as f:
    w = csv.writer(f); w.writerow(["kernel", "calls", "avg_us"] + syms)
    for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        row = [n, len(v), round(sum(v) / len(v) / 1e3, 1)]
        for s in syms:
            d = per.get((n, s)); row.append(round(sum(d.values()) / len(d), 1) if d else "")
        w.writerow(row)
        if sum(v) > 2e6: print(",".join(str(x) for x in row))
print("kernel,calls,avg_us," + ",".join(syms))
con.close()
PY
rm -rf $O/$TAG
