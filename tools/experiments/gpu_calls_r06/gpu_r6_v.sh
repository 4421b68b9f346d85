#!/bin/bash
# round 6, GPU call V: the counting pass parks the converted scalars for the placing pass (product build) against converting twice (pc8k)
mkdir -p gpurun_out/r6v; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$R/gpurun_out/r6v/scalar_cache_ab.txt
{
for round in 1 2 3; do
  for v in product pc8k; do
    for cfg in 0:1:20 0:1:n3145727 1:1:15; do
      if [ $v = product ]; then python3 tools/slice_sweep.py --quick --configs $cfg --out /tmp/ws.json > /tmp/ws.log 2>&1; else sh tools/experiments/run_with_lib.sh $v python3 tools/slice_sweep.py --quick --configs $cfg --out /tmp/ws.json > /tmp/ws.log 2>&1; fi
      [ $? = 0 ] || { echo "$v $cfg FAILED"; tail -3 /tmp/ws.log; continue; }
      python3 - "$v" "$cfg" <<'PY'
import json, sys
r = json.load(open("/tmp/ws.json"))[0]
print(f"{sys.argv[1]:8s} cfg {sys.argv[2]:13s} ok {r['ok']} total {r['total_ms']} sort {r['sort_ms']} accumulate {r['accumulate_ms']} reduce {r['reduce_ms']}")
PY
    done
  done
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_p -o kt -- python3 $R/tools/slice_sweep.py --quick --configs 0:1:20 --out /tmp/ws_p.json > /tmp/kt_p.log 2>&1
f=$(find /tmp/kt_p -name "*kernel_stats.csv" | head -1)
echo "== product: sort kernels per MSM (4 MSMs)"
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if not any(k in n for k in ("k_part", "k_bucket_pass", "k_bucket_place")): continue
    t = int(r["TotalDurationNs"]) / 1e6; c = int(r["Calls"])
    short = n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("mnt753::", "")
    print(f"   {short:50s} calls {c:4d} per MSM {t / 4:8.4f} ms")
PY
} > $O 2>&1
cat $O
cd $R
( timeout 1500 python -m pytest tests/test_msm_gpu.py -m gpu -q -x ) > gpurun_out/r6v/pytest_msm.log 2>&1; echo "pytest msm rc=$?"; tail -3 gpurun_out/r6v/pytest_msm.log | cut -c1-200
