#!/bin/bash
# round 5, GPU call N: base sets built side by side (a thread per set) -- parameter load on whatever memory the box hands over, right
# behind another prover (dirty memory), and after a pause; then the prover / full-size / live-reference suites
mkdir -p gpurun_out/r5n
export TMPDIR=/tmp
O=$PWD/gpurun_out/r5n
R=$PWD
K=/tmp/prove_keep; mkdir -p $K
python3 tools/synth_files.py MNT4753 20 $K/p4 $K/i4 > /dev/null
M=$R/snark-challenge-prover-reference_amd/main_hip
{ echo "== first prover of the box"; MNT753_TRACE=1 $M MNT4753 compute $K/p4 $K/i4 $K/o4 2>&1 | grep -i "load params\|Total time";
  echo "== a second one right behind it"; MNT753_TRACE=1 $M MNT4753 compute $K/p4 $K/i4 $K/o4 2>&1 | grep -i "load params\|Total time";
  echo "== a third right behind that"; MNT753_TRACE=1 $M MNT4753 compute $K/p4 $K/i4 $K/o4 2>&1 | grep -i "load params\|Total time";
  sleep 25; echo "== a fourth after 25 s"; MNT753_TRACE=1 $M MNT4753 compute $K/p4 $K/i4 $K/o4 --repeat 2 2>&1 | grep -i "load params\|Total time";
  echo "== unfused (five base sets), right behind"; MNT753_TRACE=1 $M MNT4753 compute $K/p4 $K/i4 $K/o4u --unfused-c 2>&1 | grep -i "load params\|Total time";
  sha256sum $K/o4 $K/o4u; } > $O/load_params_sets_side_by_side.log 2>&1
grep "^==\|^load params\|all base sets\|Total time\|o4" $O/load_params_sets_side_by_side.log
rm -rf $K
( time timeout 2400 python -m pytest tests/test_prover_gpu.py tests/test_fullsize_gpu.py tests/test_live_reference_gpu.py tests/test_groth16_gpu.py tests/test_rccl_gpu.py tests/test_bench_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-200
