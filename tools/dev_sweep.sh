#!/bin/sh
cd "$(dirname "$0")/.."
for r in 1 2 3 4 6; do echo "rounds=$r: $(MNT753_MSM_ROUNDS=$r python tools/dev_msm_big.py 20 2 | tail -1 | cut -d: -f2-)"; done
for c in 18 19 21; do echo "pre_c=$c: $(MNT753_MSM_PRE_C=$c python tools/dev_msm_big.py 20 2 | tail -1 | cut -d: -f2-)"; done
for l in 4 16 32; do echo "L=$l: $(MNT753_MSM_L=$l python tools/dev_msm_big.py 20 2 | tail -1 | cut -d: -f2-)"; done
