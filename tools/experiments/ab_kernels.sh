# GPU box: per-kernel times (rocprofv3 --kernel-trace --stats) of the G2 2^20 MSM for library variants: sh tools/experiments/ab_kernels.sh <variant> ...
R=$PWD
for v in "$@"; do
  (cd /tmp && export TMPDIR=/tmp MNT753_LIB=$R/build_exp/$v/libmnt753_hip.so && CURVE=${CURVE:-0} GROUP=${GROUP:-2} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk_$v -o x -- python3 $R/tools/dev_msm_big.py ${LOGN:-20} 3 > /dev/null 2>&1)
  echo "== $v"
  python3 - /tmp/abk_$v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:9]:
    n = r["Name"].split("(")[0].replace("void mnt753::", "")[:60]
    print(f"  {n:60s} calls {r['Calls']:>4s} avg_ms {float(r['AverageNs'])/1e6:8.3f} min_ms {float(r['MinNs'])/1e6:8.3f}")
PY
done
