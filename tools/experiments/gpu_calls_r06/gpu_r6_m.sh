#!/bin/bash
# round 6, GPU call M: the crossover of the batched-affine levels re-swept for the small sets (the rule dates from round 3; the levels
# got cheaper since): 8-way slices of 2^20, the MNT6753 sets of d = 2^15 - 1
mkdir -p gpurun_out/r6m; O=gpurun_out/r6m/levels_crossover_sweep.txt; export TMPDIR=/tmp
{
sh tools/experiments/irr_sweep.sh "0:1:17 0:1:18 0:1:19 1:1:15 1:1:n98302 0:2:17 0:2:18" "d:d 0:0 1:0 2:0 3:0 1:1 2:1 2:2 3:1 3:2"
} > $O 2>&1
cat $O
