#!/bin/bash
# round 5, GPU call K: first proof of a process with compute_H warmed at parameter load (MNT6753 2^15, MNT4753 2^20), three processes each
mkdir -p gpurun_out/r5k
export TMPDIR=/tmp
O=$PWD/gpurun_out/r5k
R=$PWD
K=/tmp/prove_keep; mkdir -p $K
python3 tools/synth_files.py MNT6753 15 $K/p6 $K/i6 > /dev/null
python3 tools/synth_files.py MNT4753 20 $K/p4 $K/i4 > /dev/null
M=$R/snark-challenge-prover-reference_amd/main_hip
for k in 1 2 3; do $M MNT6753 compute $K/p6 $K/i6 $K/o6 --repeat 3 2>&1 | grep "Total time from\|load params"; done > $O/prove6.log 2>&1; sha256sum $K/o6 >> $O/prove6.log
cat $O/prove6.log
sleep 15
for k in 1 2; do MNT753_TRACE=1 $M MNT4753 compute $K/p4 $K/i4 $K/o4 --repeat 3 2>&1 | grep "Total time from\|load params\|warm-up"; sleep 15; done > $O/prove4.log 2>&1; sha256sum $K/o4 >> $O/prove4.log
cat $O/prove4.log
( timeout 900 python -m pytest tests/test_prover_gpu.py -m gpu -q -x ) > $O/pytest_prover.log 2>&1; echo "pytest prover rc=$?"; tail -2 $O/pytest_prover.log
