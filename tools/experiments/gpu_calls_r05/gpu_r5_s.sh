#!/bin/bash
# round 5, GPU call S: the backward sweep of k_pair_level with its five products written out (-DMNT753_EXP_STRAIGHT, base fields) against
# the shipped step loop, alternating on one box: MSM parity tests on the variant, per-kernel times, bench line
mkdir -p gpurun_out/r5s
export TMPDIR=/tmp
O=$PWD/gpurun_out/r5s
R=$PWD
L=$R/build_exp/sl/libmnt753_hip.so
( MNT753_LIB=$L timeout 1200 python -m pytest tests/test_msm_gpu.py -m gpu -q -x -k "not libff" ) > $O/pytest_sl.log 2>&1
echo "pytest (straight-line variant) rc=$?"; tail -3 $O/pytest_sl.log | cut -c1-200
for round in 1 2 3; do for v in loop straight; do
  if [ $v = straight ]; then export MNT753_LIB=$L; else unset MNT753_LIB; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sl_${v}_$round -o x -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prove --no-extras --no-traffic --no-exchange > /tmp/sl_${v}_$round.json 2>/dev/null)
  python3 - /tmp/sl_${v}_$round $v $round /tmp/sl_${v}_$round.json <<'PY'
import csv, glob, sys, json
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "k_pair_level" in r["Name"] or "k_bucket_accumulate" in r["Name"]]
try:
    j = json.load(open(sys.argv[4])); extra = f"ms_per_step {j['ms_per_step']:.3f} accumulate {j['phases_ms']['accumulate_ms']:.3f} parity {j['parity_ok']}"
except Exception as ex:
    extra = "bench line: " + repr(ex)[:80]
print(f"== backward sweep: {sys.argv[2]}, round {sys.argv[3]}: {extra}")
for r in rows:
    n = r["Name"].split("(")[0].replace("void mnt753::", "").replace("mnt753::", "")[:64]
    print(f"     {n:64s} calls {r['Calls']:>3s} avg_ms {float(r['AverageNs'])/1e6:8.3f} total_ms {float(r['TotalDurationNs'])/1e6/8:8.3f} per MSM")
PY
done; done > $O/levels_straight_line_backward.txt 2>&1
unset MNT753_LIB
cat $O/levels_straight_line_backward.txt
