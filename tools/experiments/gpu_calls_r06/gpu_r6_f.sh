#!/bin/bash
# round 6, GPU call F: the final tree -- pooled buffers of the batched-affine levels (one set per device), mnt753_dev_mem_info.
# (1) the whole -m gpu suite; (2) the resident prove three times with the footprint traced; (3) the evidence bundle (tools/collect_profiles.sh)
mkdir -p gpurun_out/r6f
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6f
R=$PWD
( timeout 2400 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?"; tail -4 $O/pytest_gpu.log | cut -c1-220
D=/tmp/fp; mkdir -p $D
python3 tools/synth_files.py MNT4753 20 $D/params $D/input > /dev/null 2>&1
{ MNT753_TRACE_LOAD=1 ./snark-challenge-prover-reference_amd/main_hip MNT4753 compute $D/params $D/input $D/out --repeat 4 2>&1 | grep -E "load params|Total time|device memory"; sha256sum $D/out; } > $O/resident_prove_and_footprint.txt 2>&1
cat $O/resident_prove_and_footprint.txt
rm -rf $D
sh tools/collect_profiles.sh > $O/collect.log 2>&1
echo "collect rc=$?"; tail -3 $O/collect.log
cat gpurun_out/prof/bench_line.json | cut -c1-1500
cat gpurun_out/prof/one_shot_wall.txt
