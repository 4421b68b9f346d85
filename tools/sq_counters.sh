#!/bin/sh
# GPU box: SQ / instruction-cache counters of the MSM kernels (separate --pmc passes; kernel-trace only).  -> gpurun_out/sq/
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/sq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-prove --no-extras --no-traffic"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_IFETCH -d $O/p1 -o p1 -- python3 $R/bench.py $ARGS > $O/p1.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVES -d $O/p2 -o p2 -- python3 $R/bench.py $ARGS > $O/p2.log 2>&1
cd $R
python3 - <<'PY'
import sqlite3, glob, os, csv, collections
O = os.path.join(os.getcwd(), "gpurun_out", "sq")
def tables(cur): return [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
for db in glob.glob(O + "/*/*_results.db") + glob.glob(O + "/*/*/*_results.db"):
    con = sqlite3.connect(db); cur = con.cursor(); t = tables(cur)
    kd = [x for x in t if "kernel_dispatch" in x][0]; ks = [x for x in t if "kernel_symbol" in x][0]
    tag = os.path.basename(db).replace("_results.db", "")
    pm = [x for x in t if "pmc_event" in x]; pi = [x for x in t if "info_pmc" in x]
    if pm and pi:
        q = f"select s.display_name, p.symbol, e.value from {pm[0]} e join {pi[0]} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id"
        pa = collections.defaultdict(list)
        for n, sym, val in cur.execute(q): pa[(n.split("(")[0], sym)].append(val)
        with open(f"{O}/{tag}_pmc.csv", "w", newline="") as f:
            w = csv.writer(f); w.writerow(["kernel", "counter", "launches", "avg_value"])
            for (n, sym), v in sorted(pa.items()): w.writerow([n, sym, len(v), sum(v) / len(v)])
    con.close(); os.remove(db)
PY
ls $O
