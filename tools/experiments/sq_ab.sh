# GPU box: SQ counters of the level / accumulate kernels for library variants (build_exp/<variant>/libmnt753_hip.so), one
# rocprofv3 --pmc pass per counter group and variant:  sh tools/experiments/sq_ab.sh <variant> ...   -> gpurun_out/sq_ab/<variant>.txt
R=$PWD; O=$R/gpurun_out/sq_ab; mkdir -p $O
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-prove --no-extras --no-traffic"
for v in "$@"; do
  L=$R/build_exp/$v/libmnt753_hip.so
  (cd /tmp && export TMPDIR=/tmp MNT753_LIB=$L &&
   timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d /tmp/sqab_${v}_1 -o p -- python3 $R/bench.py $ARGS > /dev/null 2>&1
   timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD -d /tmp/sqab_${v}_2 -o p -- python3 $R/bench.py $ARGS > /dev/null 2>&1
   timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_VALU_INT64 SQ_INST_CYCLES_VMEM_WR SQ_WAVES -d /tmp/sqab_${v}_3 -o p -- python3 $R/bench.py $ARGS > /dev/null 2>&1)
  python3 - $v > $O/$v.txt <<'PY'
import sqlite3, glob, collections, sys
v = sys.argv[1]
agg = collections.defaultdict(dict)
for k in (1, 2, 3):
    for db in glob.glob(f"/tmp/sqab_{v}_{k}/**/*_results.db", recursive=True):
        con = sqlite3.connect(db); cur = con.cursor()
        t = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
        kd = [x for x in t if "kernel_dispatch" in x][0]; ks = [x for x in t if "kernel_symbol" in x][0]
        pm = [x for x in t if "pmc_event" in x][0]; pi = [x for x in t if "info_pmc" in x][0]
        q = f"select s.display_name, p.symbol, e.value, d.end - d.start from {pm} e join {pi} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id"
        acc = collections.defaultdict(list)
        for n, sym, val, dt in cur.execute(q):
            n = n.split("(")[0].replace("void mnt753::", "")
            acc[(n, sym)].append(val); acc[(n, f"us_pass{k}")].append(dt / 1e3)
        for (n, sym), vals in acc.items(): agg[n][sym] = sum(vals) / len(vals)
for n, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    if "k_pair_level" not in n and "k_bucket_accumulate" not in n: continue
    wc = d.get("SQ_WAVE_CYCLES", 1)
    print(n)
    print("   " + "  ".join(f"{k}={val:.4g}" for k, val in sorted(d.items())))
    print(f"   per wave-cycle: VALU active {d.get('SQ_ACTIVE_INST_VALU', 0)/wc:.3f}  SCA {d.get('SQ_ACTIVE_INST_SCA', 0)/wc:.3f}  LDS {d.get('SQ_ACTIVE_INST_LDS', 0)/wc:.3f}  MISC {d.get('SQ_ACTIVE_INST_MISC', 0)/wc:.3f}  "
          f"WAIT_ANY {d.get('SQ_WAIT_ANY', 0)/wc:.3f}  WAIT_INST_ANY {d.get('SQ_WAIT_INST_ANY', 0)/wc:.3f}  int64 share of VALU {d.get('SQ_INSTS_VALU_INT64', 0)/max(d.get('SQ_INSTS_VALU', 1), 1):.3f}  "
          f"icache miss rate {d.get('SQC_ICACHE_MISSES', 0)/max(d.get('SQC_ICACHE_REQ', 1), 1):.4f}")
PY
  echo "== $v"; cat $O/$v.txt
done
