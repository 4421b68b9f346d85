#!/bin/bash
# round 5, GPU call I: the proves after the host-tail changes (binary-Euclid inversion, results touched in completion order)
mkdir -p gpurun_out/r5i
export TMPDIR=/tmp
O=$PWD/gpurun_out/r5i
R=$PWD
K=/tmp/prove_keep; mkdir -p $K
python3 tools/synth_files.py MNT6753 15 $K/p6 $K/i6 > /dev/null
python3 tools/synth_files.py MNT4753 20 $K/p4 $K/i4 > /dev/null
M=$R/snark-challenge-prover-reference_amd/main_hip
for k in 1 2; do $M MNT6753 compute $K/p6 $K/i6 $K/o6 --repeat 6 2>&1 | grep "Total time from\|store\|gpu:"; sha256sum $K/o6; done > $O/prove6.log 2>&1
cat $O/prove6.log | grep "Total\|o6" 
$M MNT4753 compute $K/p4 $K/i4 $K/o4 --repeat 4 > $O/prove4.log 2>&1; sha256sum $K/o4 >> $O/prove4.log
grep "Total time from\|load params\|o4" $O/prove4.log
grep -A3 "MNT4753.*20\|MNT6753.*15" tests/golden/oracle_hashes.json | head -12
( timeout 1500 python -m pytest tests/test_prover_gpu.py tests/test_live_reference_gpu.py tests/test_groth16_gpu.py -m gpu -q -x ) > $O/pytest_prover.log 2>&1; echo "pytest prover rc=$?"; tail -3 $O/pytest_prover.log
