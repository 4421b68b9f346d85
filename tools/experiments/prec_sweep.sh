#!/bin/sh
# GPU box: G1 / G2 MSM time by window width of the precomputed table (MNT753_MSM_PRE_C) at 2^20 and 2^21 points.
#   sh tools/experiments/prec_sweep.sh > gpurun_out/prec_sweep.txt
for cfg in ${PREC_CONFIGS:-0:1:20 0:1:21 0:2:20}; do
  for c in ${PREC_CS:-19 20 21 22 23}; do
    MNT753_MSM_PRE_C=$c python3 tools/slice_sweep.py --quick --configs $cfg --out /tmp/prec_$c.json > /dev/null 2>&1
    python3 - "$cfg" "$c" /tmp/prec_$c.json <<'PY'
import json, sys
r = json.load(open(sys.argv[3]))[0]
print(f"cfg {sys.argv[1]} pre_c {sys.argv[2]}: window_bits {r['window_bits']} levels {r['levels']} ok {r['ok']} total {r['total_ms']} sort {r['sort_ms']} accumulate {r['accumulate_ms']} reduce {r['reduce_ms']}")
PY
  done
done
