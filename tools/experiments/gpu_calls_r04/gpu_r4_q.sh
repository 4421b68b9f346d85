#!/bin/bash
# round 4, GPU call Q: host small-constant products as additions (parity), pairing levels for mid-size base-field sets
mkdir -p gpurun_out/r4q
O=$PWD/gpurun_out/r4q
( time python -m pytest tests/test_msm_gpu.py tests/test_prover_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^FAILED|passed|failed" $O/pytest.log | cut -c1-150 | head
python tools/experiments/pair_levels_small_sweep.py > $O/pair_levels_small_sweep.txt 2>&1; echo "sweep rc=$?"
cut -c1-230 $O/pair_levels_small_sweep.txt
