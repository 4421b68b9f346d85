#!/bin/bash
# round 6, GPU call C: (1) the self-test, one-shot and prover tests on the product as built (G1 levels: final step loop; lane-split
# levels in the same loop shape); (2) G1 A/B: where the prefix product is read in the later levels (pt2) and two DMA portions there (d2);
# (3) G2 A/B: the lane-split loop against the round-5 loop (g2old); (4) wall clock of the one-shot prover against --tables; (5) self-test time
mkdir -p gpurun_out/r6c
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6c
R=$PWD
( timeout 1500 python -m pytest tests/test_selftest_gpu.py tests/test_prover_gpu.py tests/test_msm_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-220
kstats() {  # $1 = label, $2 = lib or "", $3.. = command
  local label=$1 lib=$2; shift 2
  if [ -n "$lib" ]; then export MNT753_LIB=$lib; else unset MNT753_LIB; fi
  (cd /tmp && rm -rf /tmp/ks_$label && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$label -o x -- "$@" > /tmp/ks_$label.out 2>/dev/null)
  unset MNT753_LIB
  python3 - /tmp/ks_$label $label <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "k_pair_level" in r["Name"] or "k_bucket_accumulate" in r["Name"]]
tot = 0.0
print(f"== {sys.argv[2]}")
for r in rows:
    n = r["Name"].split("(")[0].replace("void mnt753::", "").replace("mnt753::", "")[:60]
    print(f"     {n:60s} calls {r['Calls']:>3s} avg_ms {float(r['AverageNs'])/1e6:8.3f}")
PY
}
{
for round in 1 2; do
  kstats g1_product_$round "" python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prove --no-extras --no-traffic --no-exchange
  for v in pt2 d2; do [ -f $R/build_exp/$v/libmnt753_hip.so ] && kstats g1_${v}_$round $R/build_exp/$v/libmnt753_hip.so python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prove --no-extras --no-traffic --no-exchange; done
done
} > $O/g1_ab.txt 2>&1
cat $O/g1_ab.txt
{
for round in 1 2; do
  CURVE=0 GROUP=2 kstats g2_product_$round "" python3 $R/tools/dev_msm_big.py 20 3
  [ -f $R/build_exp/g2old/libmnt753_hip.so ] && CURVE=0 GROUP=2 kstats g2_old_$round $R/build_exp/g2old/libmnt753_hip.so python3 $R/tools/dev_msm_big.py 20 3
done
CURVE=1 GROUP=2 kstats m6g2_product "" python3 $R/tools/dev_msm_big.py 15 3
[ -f $R/build_exp/g2old/libmnt753_hip.so ] && CURVE=1 GROUP=2 kstats m6g2_old $R/build_exp/g2old/libmnt753_hip.so python3 $R/tools/dev_msm_big.py 15 3
} > $O/g2_ab.txt 2>&1
cat $O/g2_ab.txt
# untimed totals, alternating (G2 2^20 through dev_msm_big, G1 through bench.py)
for round in 1 2 3; do for v in product g2old; do
  if [ $v = g2old ]; then export MNT753_LIB=$R/build_exp/g2old/libmnt753_hip.so; [ -f $MNT753_LIB ] || continue; else unset MNT753_LIB; fi
  g2=$(CURVE=0 GROUP=2 timeout 300 python3 tools/dev_msm_big.py 20 4 2>/dev/null | tail -1 | sed 's/.*total_ms=//')
  m6=$(CURVE=1 GROUP=2 timeout 300 python3 tools/dev_msm_big.py 15 4 2>/dev/null | tail -1 | sed 's/.*total_ms=//')
  echo "round $round $v: MNT4753 G2 2^20 $g2 | MNT6753 G2 2^15 $m6"
done; done > $O/g2_totals.txt 2>&1
unset MNT753_LIB
cat $O/g2_totals.txt
# (4) one-shot prover: wall clock of the process
D=/tmp/oneshot; mkdir -p $D
python3 tools/synth_files.py MNT4753 20 $D/params $D/input > /dev/null 2>&1
{
for k in 1 2 3; do
  sleep 15
  /usr/bin/time -f "one-shot   wall %e s" ./snark-challenge-prover-reference_amd/main_hip MNT4753 compute $D/params $D/input $D/out1 2>&1 | grep -E "wall|load params|Total time|one-shot"
  sleep 15
  /usr/bin/time -f "--tables   wall %e s" ./snark-challenge-prover-reference_amd/main_hip MNT4753 compute $D/params $D/input $D/out2 --tables 2>&1 | grep -E "wall|load params|Total time"
  cmp $D/out1 $D/out2 && echo "same bytes"
done
MNT753_TRACE_LOAD=1 ./snark-challenge-prover-reference_amd/main_hip MNT4753 compute $D/params $D/input $D/out1 2>&1 | grep -E "load params|self-test|one-shot|Total"
sha256sum $D/out1
} > $O/one_shot_wall.txt 2>&1
cat $O/one_shot_wall.txt
rm -rf $D
# (5) self-test time
python3 - <<'PY' > $O/self_test_time.txt 2>&1
import time, sys, os
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
pkg = load_package(); pkg.init(0)
for level in (0, 1, 1, 1, 2, 2):
    t = time.time(); pkg.self_test(level); print(f"mnt753_self_test({level}): {1e3 * (time.time() - t):.1f} ms")
PY
cat $O/self_test_time.txt
