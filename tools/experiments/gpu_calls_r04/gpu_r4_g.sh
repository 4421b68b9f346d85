#!/bin/bash
# round 4, GPU call G: which golden MSMs fail with the tree merge, under which lane share
mkdir -p gpurun_out/r4g
O=$PWD/gpurun_out/r4g
for env in "X=1" "MNT753_MSM_TMIN=16" "MNT753_MSM_TMIN=4" "MNT753_MSM_TMIN=1" "MNT753_EDGE_TREE=0" "MNT753_EDGE_TREE=0 MNT753_MSM_TMIN=1"; do
  echo "== $env"
  env $env python -m pytest tests/test_msm_gpu.py -m gpu -q -k "test_golden" 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-120
done > $O/golden_matrix.log 2>&1
cat $O/golden_matrix.log
