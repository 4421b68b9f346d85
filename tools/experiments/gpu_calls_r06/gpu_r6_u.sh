#!/bin/bash
# round 6, GPU call U: chunk of the level-2 placing pass, 16384 pairs (one workgroup per CU) against 8192 (two)
mkdir -p gpurun_out/r6u; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$R/gpurun_out/r6u/place_chunk_ab.txt
{
for round in 1 2 3; do
  for v in pc16k pc8k; do
    for cfg in 0:1:20 0:1:n3145727; do
      sh tools/experiments/run_with_lib.sh $v python3 tools/slice_sweep.py --quick --configs $cfg --out /tmp/ws.json > /tmp/ws.log 2>&1 || { echo "$v $cfg FAILED"; tail -3 /tmp/ws.log; continue; }
      python3 - "$v" "$cfg" <<'PY'
import json, sys
r = json.load(open("/tmp/ws.json"))[0]
print(f"{sys.argv[1]:7s} cfg {sys.argv[2]:13s} ok {r['ok']} total {r['total_ms']} sort {r['sort_ms']} accumulate {r['accumulate_ms']} reduce {r['reduce_ms']}")
PY
    done
  done
done
cd /tmp
for v in pc16k pc8k; do
  export MNT753_LIB=$R/build_exp/$v/libmnt753_hip.so
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -o kt -- python3 $R/tools/slice_sweep.py --quick --configs 0:1:20 --out /tmp/ws_$v.json > /tmp/kt_$v.log 2>&1
  f=$(find /tmp/kt_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v: sort kernels per MSM (4 MSMs)"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if not any(k in n for k in ("k_part", "k_bucket_pass", "k_bucket_place", "k_pair_level<mnt753::Mnt4G1, true")): continue
    t = int(r["TotalDurationNs"]) / 1e6; c = int(r["Calls"])
    short = n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("mnt753::", "")
    print(f"   {short:50s} calls {c:4d} per MSM {t / 4:8.4f} ms")
PY
done
} > $O 2>&1
cat $O
