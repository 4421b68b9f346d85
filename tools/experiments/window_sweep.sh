#!/bin/sh
# GPU box: MSM time by window width of the table (MNT753_MSM_TABLE_BITS) and level counts:
#   sh tools/experiments/window_sweep.sh "<curve:group:size> ..." "<bits> ..." "<R:K> ..."
# (the rule the product follows: pick_precomp_bits / pair_levels / irr_levels_for, csrc/msm_host.hpp; "d" = that rule)
for cfg in ${1:-0:1:20 0:1:n3145727 0:2:20}; do
  for bits in ${2:-d 18 19 20 21}; do
    if [ "$bits" = d ]; then unset MNT753_MSM_TABLE_BITS; else export MNT753_MSM_TABLE_BITS=$bits; fi
    for rk in ${3:-d:d}; do
      R=${rk%%:*}; K=${rk##*:}
      if [ "$K" = d ]; then unset MNT753_MSM_IRR; else export MNT753_MSM_IRR=$K; fi
      if [ "$R" = d ]; then unset MNT753_MSM_PAIR; else export MNT753_MSM_PAIR=$R; fi
      timeout 300 python3 tools/slice_sweep.py --quick --configs $cfg --out /tmp/ws.json > /tmp/ws.log 2>&1 || { echo "cfg $cfg bits $bits R $R K $K: FAILED"; tail -3 /tmp/ws.log; continue; }
      python3 - "$cfg" "$bits" "$R" "$K" /tmp/ws.json <<'PY'
import json, sys
r = json.load(open(sys.argv[5]))[0]
print(f"cfg {sys.argv[1]} table bits {sys.argv[2]} (plan c {r.get('window_bits')}) regular {sys.argv[3]} irregular {sys.argv[4]} (ran {r.get('levels')} + {r.get('irr_levels')}): ok {r['ok']} total {r['total_ms']} sort {r['sort_ms']} accumulate {r['accumulate_ms']} reduce {r['reduce_ms']}")
PY
    done
  done
done
