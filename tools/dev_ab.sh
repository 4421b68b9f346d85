#!/bin/sh
# A/B: wave-uniform accumulate kernel vs lane-divergent VM kernel, same binary, same data
cd "$(dirname "$0")/.."
for mode in uniform vm; do
  echo "== MNT753_MSM_ACC=$mode"
  MNT753_MSM_ACC=$mode timeout 300 python tools/dev_msm_check.py all | tail -1
  MNT753_MSM_ACC=$mode MNT753_MSM_PRECOMP=1 timeout 300 python tools/dev_msm_check.py all | tail -1
  MNT753_MSM_ACC=$mode timeout 600 python tools/dev_msm_big.py 20 3 | tail -2
  MNT753_MSM_ACC=$mode CURVE=0 GROUP=2 timeout 600 python tools/dev_msm_big.py 18 2 | tail -1
done
