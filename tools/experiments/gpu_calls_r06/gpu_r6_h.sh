#!/bin/bash
# round 6, GPU call H: with the level kernels 5-8 % cheaper, do the level counts still sit right?  G1 2^20 (A's MSM) and G1 3 * 2^20 - 2
# points (C's) by (regular, irregular) levels; `d` = the rule in csrc/msm_host.hpp (pair_levels, irr_levels_for)
mkdir -p gpurun_out/r6h
export TMPDIR=/tmp
sh tools/experiments/irr_sweep.sh "0:1:20 0:1:n3145727" "d:d 3:2 3:3 3:4 4:1 4:2 2:3" > gpurun_out/r6h/level_counts_sweep.txt 2>&1
cat gpurun_out/r6h/level_counts_sweep.txt
sh tools/experiments/irr_sweep.sh "0:1:20 0:1:n3145727" "d:d 3:3" >> gpurun_out/r6h/level_counts_sweep.txt 2>&1
tail -4 gpurun_out/r6h/level_counts_sweep.txt
