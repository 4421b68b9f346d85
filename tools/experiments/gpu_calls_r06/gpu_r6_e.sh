#!/bin/bash
# round 6, GPU call E: (1) the WHOLE -m gpu suite on the product as built; (2) G2 A/B of the DMA portions of the lane-split later levels:
# product = 3 (regular and irregular) against g2l2 = 2 regular, g2l2i2 = 2 regular and irregular; MNT4753 G2 2^20 and MNT6753 G2 2^15
mkdir -p gpurun_out/r6e
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6e
R=$PWD
( timeout 2400 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?"; tail -6 $O/pytest_gpu.log | cut -c1-220
kstats() {
  local label=$1 lib=$2; shift 2
  if [ -n "$lib" ]; then export MNT753_LIB=$lib; else unset MNT753_LIB; fi
  (cd /tmp && rm -rf /tmp/ks_$label && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$label -o x -- "$@" > /tmp/ks_$label.out 2>/dev/null)
  unset MNT753_LIB
  python3 - /tmp/ks_$label $label /tmp/ks_$label.out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "k_pair_level" in r["Name"] or "k_bucket_accumulate" in r["Name"]]
last = [l.strip() for l in open(sys.argv[3]) if "total_ms" in l][-1:]
print(f"== {sys.argv[2]}  {last[0][last[0].find('wall'):] if last else ''}")
for r in rows:
    n = r["Name"].split("(")[0].replace("void mnt753::", "").replace("mnt753::", "")[:60]
    print(f"     {n:60s} calls {r['Calls']:>3s} avg_ms {float(r['AverageNs'])/1e6:8.3f}")
PY
}
{
for round in 1 2; do
  for v in product g2l2 g2l2i2; do
    lib=""; [ $v != product ] && lib=$R/build_exp/$v/libmnt753_hip.so
    CURVE=0 GROUP=2 kstats g2_${v}_$round "$lib" python3 $R/tools/dev_msm_big.py 20 3
    CURVE=1 GROUP=2 kstats m6g2_${v}_$round "$lib" python3 $R/tools/dev_msm_big.py 15 3
  done
done
} > $O/g2_dma_portions_ab.txt 2>&1
cat $O/g2_dma_portions_ab.txt
