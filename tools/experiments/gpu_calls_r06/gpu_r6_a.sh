#!/bin/bash
# round 6, GPU call A: the backward sweep of k_pair_level with its operands in place (MNT753_PAIR_DIET=1, the product as built) against
# the step loop of round 5 (build_exp/diet0: -DMNT753_PAIR_DIET=0), alternating on one box: MSM parity tests on the product,
# per-kernel times, SQ counters (VALU instructions, busy cycles = clock) of both.
mkdir -p gpurun_out/r6a
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6a
R=$PWD
L0=$R/build_exp/diet0/libmnt753_hip.so
( timeout 1500 python -m pytest tests/test_msm_gpu.py tests/test_device_kat_gpu.py -m gpu -q -x ) > $O/pytest_diet.log 2>&1
echo "pytest (diet product) rc=$?"; tail -3 $O/pytest_diet.log | cut -c1-200
for round in 1 2 3; do for v in r5loop diet; do
  if [ $v = r5loop ]; then export MNT753_LIB=$L0; else unset MNT753_LIB; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/d_${v}_$round -o x -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prove --no-extras --no-traffic --no-exchange > /tmp/d_${v}_$round.json 2>/dev/null)
  python3 - /tmp/d_${v}_$round $v $round /tmp/d_${v}_$round.json <<'PY'
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
done; done > $O/levels_operands_in_place.txt 2>&1
unset MNT753_LIB
cat $O/levels_operands_in_place.txt
# untraced timing, alternating
for round in 1 2 3; do for v in r5loop diet; do
  if [ $v = r5loop ]; then export MNT753_LIB=$L0; else unset MNT753_LIB; fi
  timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prove --no-extras --no-traffic --no-exchange 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$v round $round', round(j['ms_per_step'],3), {k: round(v,3) for k,v in j['phases_ms'].items()}, j['parity_ok'])"
done; done > $O/bench_ab.txt 2>&1
unset MNT753_LIB
cat $O/bench_ab.txt
# SQ counters of both
mkdir -p build_exp/diet1 && cp snark-challenge-prover-reference_amd/libmnt753_hip.so build_exp/diet1/
sh tools/experiments/sq_ab.sh diet0 diet1 > $O/sq_ab.log 2>&1
cp gpurun_out/sq_ab/diet0.txt $O/sq_levels_r5_step_loop.txt; cp gpurun_out/sq_ab/diet1.txt $O/sq_levels_operands_in_place.txt
cat $O/sq_levels_r5_step_loop.txt $O/sq_levels_operands_in_place.txt | grep -v "^   SQ" | cut -c1-260
