#!/bin/bash
# round 5, GPU call Q: call G again with the prefetch of k_pair_level1w repaired (operands raw until their slot, loads unconditional:
# the first form waited for its loads where it issued them)
mkdir -p gpurun_out/r5r
export TMPDIR=/tmp
O=$PWD/gpurun_out/r5r
R=$PWD
L=$R/build_exp/p2w/libmnt753_hip.so
( MNT753_LIB=$L MNT753_EXP_PAIR2W=2 timeout 900 python -m pytest tests/test_msm_gpu.py -m gpu -q -x -k "skewed or cancellation or pairing or randomized or full_size" ) > $O/pytest_p1w.log 2>&1
echo "pytest (1 wave, no image) rc=$?"; tail -3 $O/pytest_p1w.log | cut -c1-200
for round in 1 2 3; do for v in image twowave prefetch; do
  case $v in image) unset MNT753_EXP_PAIR2W;; twowave) export MNT753_EXP_PAIR2W=1;; prefetch) export MNT753_EXP_PAIR2W=2;; esac
  (cd /tmp && MNT753_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1w_${v}_$round -o x -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prove --no-extras --no-traffic --no-exchange > /tmp/p1w_${v}_$round.json 2>/dev/null)
  python3 - /tmp/p1w_${v}_$round $v $round /tmp/p1w_${v}_$round.json <<'PY'
import csv, glob, sys, json
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "k_pair_level" in r["Name"] or "k_bucket_accumulate" in r["Name"]]
try:
    j = json.load(open(sys.argv[4])); extra = f"ms_per_step {j['ms_per_step']:.3f} accumulate {j['phases_ms']['accumulate_ms']:.3f} parity {j['parity_ok']}"
except Exception as ex:
    extra = "bench line: " + repr(ex)[:80]
print(f"== levels behind the first: {sys.argv[2]}, round {sys.argv[3]}: {extra}")
for r in rows:
    n = r["Name"].split("(")[0].replace("void mnt753::", "").replace("mnt753::", "")[:64]
    print(f"     {n:64s} calls {r['Calls']:>3s} avg_ms {float(r['AverageNs'])/1e6:8.3f} total_ms {float(r['TotalDurationNs'])/1e6/8:8.3f} per MSM")
PY
done; done > $O/levels_no_image_register_prefetch_v3.txt 2>&1
grep "==\|1w\|2w\|false, false, false\|false, false, true\|false, true, true" $O/levels_no_image_register_prefetch_v3.txt
export MNT753_EXP_PAIR2W=2
sh tools/experiments/sq_ab.sh p2w > $O/sq_one_wave_prefetch_v3.txt 2>&1
unset MNT753_EXP_PAIR2W
grep -A2 "k_pair_level1w<mnt753::Mnt4G1, false, false>" $O/sq_one_wave_prefetch_v3.txt | cut -c1-1200
