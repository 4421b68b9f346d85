#!/bin/sh
# GPU box: MSM time by (regular levels, irregular levels):  sh tools/experiments/irr_sweep.sh "<curve:group:size> ..." "<R:K> ..."
for cfg in ${1:-0:1:20 0:2:20}; do
  for rk in ${2:-3:0 3:1 3:2 3:3 2:2 2:3 1:3 1:4}; do
    R=${rk%%:*}; K=${rk##*:}
    if [ "$K" = d ]; then unset MNT753_MSM_IRR; else export MNT753_MSM_IRR=$K; fi   # d: the default rule (irr_levels_for, csrc/msm_host.hpp)
    if [ "$R" = d ]; then unset MNT753_MSM_PAIR; else export MNT753_MSM_PAIR=$R; fi
    timeout 300 python3 tools/slice_sweep.py --quick --configs $cfg --out /tmp/irr_$R$K.json > /tmp/irr_$R$K.log 2>&1 || { echo "cfg $cfg R $R K $K: FAILED"; tail -3 /tmp/irr_$R$K.log; continue; }
    python3 - "$cfg" "$R" "$K" /tmp/irr_$R$K.json <<'PY'
import json, sys
r = json.load(open(sys.argv[4]))[0]
print(f"cfg {sys.argv[1]} regular {sys.argv[2]} irregular {sys.argv[3]} (ran {r.get('levels')} + {r.get('irr_levels')}): ok {r['ok']} total {r['total_ms']} sort {r['sort_ms']} accumulate {r['accumulate_ms']} reduce {r['reduce_ms']}")
PY
  done
done
