#!/bin/bash
# round 6, GPU call K: which kernels of the accumulation phase do not follow the entry count when the table's window grows (c = 19 -> 20)
mkdir -p gpurun_out/r6k; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for bits in 18 19 20 21; do
  export MNT753_MSM_TABLE_BITS=$bits
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$bits -o kt -- python3 $R/tools/slice_sweep.py --quick --configs 0:1:20 --out /tmp/ws_$bits.json > /tmp/kt_$bits.log 2>&1
  f=$(find /tmp/kt_$bits -name "*kernel_stats.csv" | head -1); [ -z "$f" ] && { find /tmp/kt_$bits | head; tail -5 /tmp/kt_$bits.log; }
  echo "== table bits $bits"; grep -h '"ok"' /tmp/kt_$bits.log | head -1 | cut -c1-300
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "precompute" in n or "synth" in n: continue
    t = int(r["TotalDurationNs"]) / 1e6; c = int(r["Calls"])
    if t / 4 < 0.05: continue
    short = n.split("(")[0].replace("void ", "").replace("mnt753::", "").replace("(anonymous namespace)::", "")
    print(f"   {short:60s} calls {c:4d} per MSM (4 MSMs) {t / 4:8.3f} ms")
PY
done > $R/gpurun_out/r6k/kernels_by_window_width.txt 2>&1
cat $R/gpurun_out/r6k/kernels_by_window_width.txt
