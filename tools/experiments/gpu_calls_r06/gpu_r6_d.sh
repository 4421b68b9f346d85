#!/bin/bash
# round 6, GPU call D: (1) tests touched since call C; (2) G1 A/B of the DMA portions per kind of level: product = (first 4, later 2,
# irregular 3) against da = later 1, db = first 3, dc = irregular 4; (3) the one-shot prover's wall clock (no tables, no level buffers,
# no warm-up) against --tables, three times each with pauses; (4) self-test time per curve; (5) MNT6753 prove (k_scalar_digits probe)
mkdir -p gpurun_out/r6d
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6d
R=$PWD
( timeout 1200 python -m pytest tests/test_selftest_gpu.py tests/test_prover_gpu.py tests/test_msm_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-220
kstats() {
  local label=$1 lib=$2; shift 2
  if [ -n "$lib" ]; then export MNT753_LIB=$lib; else unset MNT753_LIB; fi
  (cd /tmp && rm -rf /tmp/ks_$label && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$label -o x -- "$@" > /tmp/ks_$label.out 2>/dev/null)
  unset MNT753_LIB
  python3 - /tmp/ks_$label $label /tmp/ks_$label.out <<'PY'
import csv, glob, sys, json
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "k_pair_level" in r["Name"] or "k_bucket_accumulate" in r["Name"]]
try:
    j = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1]); extra = f"ms_per_step {j['ms_per_step']:.3f} accumulate {j['phases_ms']['accumulate_ms']:.3f} parity {j['parity_ok']}"
except Exception as ex:
    extra = ""
print(f"== {sys.argv[2]} {extra}")
for r in rows:
    n = r["Name"].split("(")[0].replace("void mnt753::", "").replace("mnt753::", "")[:60]
    print(f"     {n:60s} calls {r['Calls']:>3s} avg_ms {float(r['AverageNs'])/1e6:8.3f}")
PY
}
B="python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prove --no-extras --no-traffic --no-exchange"
{
for round in 1 2; do
  kstats g1_product_$round "" $B
  for v in da db dc; do [ -f $R/build_exp/$v/libmnt753_hip.so ] && kstats g1_${v}_$round $R/build_exp/$v/libmnt753_hip.so $B; done
done
} > $O/g1_dma_portions_ab2.txt 2>&1
cat $O/g1_dma_portions_ab2.txt
# (3) one-shot wall
D=/tmp/oneshot; mkdir -p $D
python3 tools/synth_files.py MNT4753 20 $D/params $D/input > /dev/null 2>&1
M=./snark-challenge-prover-reference_amd/main_hip
wall() { local t0=$(date +%s.%N); "$@" > $D/stdout.txt 2> $D/stderr.txt; local rc=$?; local t1=$(date +%s.%N); echo "rc $rc wall $(python3 -c "print(round($t1 - $t0, 3))") s | $(grep -E 'load params:|Total time from' $D/stdout.txt | tr '\n' ' ')"; }
{
for k in 1 2 3; do
  sleep 20; echo -n "one-shot  : "; wall $M MNT4753 compute $D/params $D/input $D/out1
  sleep 20; echo -n "--tables  : "; wall $M MNT4753 compute $D/params $D/input $D/out2 --tables
  cmp $D/out1 $D/out2 && echo "same bytes"
done
sleep 20
MNT753_TRACE_LOAD=1 $M MNT4753 compute $D/params $D/input $D/out1 2>&1 | grep -E "load params|self-test|one-shot|Total"
sha256sum $D/out1; grep -A1 MNT4753_2p20 tests/golden/oracle_hashes.json | head -3; grep output_sha256 tests/golden/oracle_hashes.json | head -4
} > $O/one_shot_wall.txt 2>&1
cat $O/one_shot_wall.txt
rm -rf $D
# (4) self-test time
python3 - <<'PY' > $O/self_test_time.txt 2>&1
import time, sys, os
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
pkg = load_package(); pkg.init(0)
for level, curve in ((0, None), (1, 0), (1, 0), (1, 1), (1, 1), (1, None), (2, None)):
    t = time.time(); pkg.self_test(level, curve=curve); print(f"mnt753_self_test(level {level}, curve {curve}): {1e3 * (time.time() - t):.1f} ms")
PY
cat $O/self_test_time.txt
# (5) MNT6753 prove, resident
D=/tmp/m6; mkdir -p $D
python3 tools/synth_files.py MNT6753 15 $D/params $D/input > /dev/null 2>&1
{ $M MNT6753 compute $D/params $D/input $D/out --repeat 6 | grep -E "Total time|load params"; sha256sum $D/out; grep -B2 -A6 MNT6753_2p15 tests/golden/oracle_hashes.json | grep output_sha256; } > $O/mnt6753_prove.txt 2>&1
cat $O/mnt6753_prove.txt
(cd /tmp && CURVE=1 GROUP=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_m6g1 -o x -- python3 $R/tools/dev_msm_big.py 15 3 > /dev/null 2>&1; python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/ks_m6g1/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:12]:
    print(f"  {r['Name'].split('(')[0][-60:]:60s} calls {r['Calls']:>4s} avg_ms {float(r['AverageNs'])/1e6:8.4f}")
PY
) > $O/mnt6_g1_2p15_kernels.txt 2>&1
cat $O/mnt6_g1_2p15_kernels.txt
rm -rf $D
