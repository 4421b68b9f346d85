#!/bin/bash
# round 4, GPU call M: window walk split of k_scalar_digits, wider list-level bounds, floor of entries per lane
mkdir -p gpurun_out/r4m
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4m
R=$PWD
( time python -m pytest tests/test_device_kat_gpu.py tests/test_msm_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^FAILED|passed|failed" $O/pytest.log | cut -c1-150 | head -20
python tools/experiments/tmin_sweep.py > $O/tmin_sweep.txt 2>&1; echo "sweep rc=$?"
cut -c1-220 $O/tmin_sweep.txt
