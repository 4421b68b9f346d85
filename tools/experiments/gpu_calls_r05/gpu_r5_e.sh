#!/bin/bash
# round 5, GPU call E: SQ counters of the level kernels at one and at two waves per SIMD (tools/experiments/sq_ab.sh on the p2w build)
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/r5e
unset MNT753_EXP_PAIR2W
sh tools/experiments/sq_ab.sh p2w > gpurun_out/r5e/sq_one_wave.txt 2>&1
export MNT753_EXP_PAIR2W=1
sh tools/experiments/sq_ab.sh p2w > gpurun_out/r5e/sq_two_waves.txt 2>&1
unset MNT753_EXP_PAIR2W
grep -A2 "k_pair_level<mnt753::Mnt4G1, false, false, false>\|k_pair_level2w<mnt753::Mnt4G1, false, false>" gpurun_out/r5e/sq_one_wave.txt gpurun_out/r5e/sq_two_waves.txt | cut -c1-1200
