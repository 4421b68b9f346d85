#!/bin/bash
# round 6, GPU call J: window width of the table re-swept behind the cheaper levels (the rule dates from round 3)
mkdir -p gpurun_out/r6j; O=gpurun_out/r6j/window_width_sweep.txt
export TMPDIR=/tmp
{
sh tools/experiments/window_sweep.sh "0:1:20" "18 21" "d:d"
sh tools/experiments/window_sweep.sh "0:1:20" "d 19 20" "d:d 4:2 3:3"
sh tools/experiments/window_sweep.sh "0:1:n3145727" "d 20 21 22" "d:d"
sh tools/experiments/window_sweep.sh "0:1:n3145727" "21 22" "4:2 4:3"
sh tools/experiments/window_sweep.sh "0:2:20" "d 18 19 20" "d:d"
sh tools/experiments/window_sweep.sh "0:1:20" "d 20" "d:d"
} > $O 2>&1
cat $O
