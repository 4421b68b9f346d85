#!/bin/bash
# round 6, GPU call R: the partition passes of the sort stage with the window width as a template parameter (k_part_pass_c) against
# the generic kernels (MNT753_MSM_SORT=generic), alternating; parity through the MSM tests
mkdir -p gpurun_out/r6r; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$R/gpurun_out/r6r/sort_by_width_ab.txt
{
for round in 1 2 3; do
  for v in by_width generic; do
    if [ $v = generic ]; then export MNT753_MSM_SORT=generic; else unset MNT753_MSM_SORT; fi
    for cfg in 0:1:20 0:1:n3145727 0:2:20 1:1:15 1:2:15; do
      python3 tools/slice_sweep.py --quick --configs $cfg --out /tmp/ws.json > /tmp/ws.log 2>&1 || { echo "$v $cfg FAILED"; tail -3 /tmp/ws.log; continue; }
      python3 - "$v" "$cfg" <<'PY'
import json, sys
r = json.load(open("/tmp/ws.json"))[0]
print(f"round {sys.argv[1]:9s} cfg {sys.argv[2]:13s} c {r['window_bits']} ok {r['ok']} total {r['total_ms']} sort {r['sort_ms']} accumulate {r['accumulate_ms']} reduce {r['reduce_ms']}")
PY
    done
  done
done
unset MNT753_MSM_SORT
cd /tmp
for v in by_width generic; do
  if [ $v = generic ]; then export MNT753_MSM_SORT=generic; else unset MNT753_MSM_SORT; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -o kt -- python3 $R/tools/slice_sweep.py --quick --configs 0:1:20 --out /tmp/ws_$v.json > /tmp/kt_$v.log 2>&1
  f=$(find /tmp/kt_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v: sort kernels per MSM (4 MSMs)"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if not any(k in n for k in ("k_part", "k_bucket_pass", "k_bucket_place", "k_bucket_pad", "k_scan")): continue
    t = int(r["TotalDurationNs"]) / 1e6; c = int(r["Calls"])
    short = n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("mnt753::", "")
    print(f"   {short:50s} calls {c:4d} per MSM {t / 4:8.4f} ms")
PY
done
} > $O 2>&1
cat $O
cd $R; unset MNT753_MSM_SORT
( timeout 1500 python -m pytest tests/test_msm_gpu.py tests/test_selftest_gpu.py -m gpu -q -x ) > gpurun_out/r6r/pytest_msm.log 2>&1; echo "pytest msm rc=$?"; tail -3 gpurun_out/r6r/pytest_msm.log | cut -c1-200
