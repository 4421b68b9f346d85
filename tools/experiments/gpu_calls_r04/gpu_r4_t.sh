#!/bin/bash
# round 4, GPU call T: the MSM and prover suites under every switch of the merge / the lane groups (the non-default paths stay selectable)
mkdir -p gpurun_out/r4t
O=$PWD/gpurun_out/r4t
for env in "MNT753_FLOW=0" "MNT753_EDGE_TREE=0" "MNT753_EDGE_FLOW_NODES=0" "MNT753_EDGE_FLOW_NODES=100000000 MNT753_REDUCE_FLOW_MAX=100000000" "MNT753_REDUCE_FLOW_MAX=0 MNT753_MSM_TMIN=2"; do
  echo "== $env"
  env $env python -m pytest tests/test_msm_gpu.py tests/test_prover_gpu.py -m gpu -q 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-160
done > $O/switch_matrix.log 2>&1
cat $O/switch_matrix.log
