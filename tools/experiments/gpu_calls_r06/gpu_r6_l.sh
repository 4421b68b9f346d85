#!/bin/bash
# round 6, GPU call L: the kernels of one resident proof on a time axis (MNT6753 2^15 and MNT4753 2^20): where does the device idle?
mkdir -p gpurun_out/r6l; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
K=/tmp/pk; mkdir -p $K
python3 tools/synth_files.py MNT6753 15 $K/p6 $K/i6 > /dev/null
python3 tools/synth_files.py MNT4753 20 $K/p4 $K/i4 > /dev/null
M=$R/snark-challenge-prover-reference_amd/main_hip
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl6 -o tl -- $M MNT6753 compute $K/p6 $K/i6 $K/o6 --repeat 4 > $R/gpurun_out/r6l/m6_stdout.txt 2>&1
python3 $R/tools/prove_timeline.py /tmp/tl6 --gap-ms 0.4 --all > $R/gpurun_out/r6l/timeline_mnt6753.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl4 -o tl -- $M MNT4753 compute $K/p4 $K/i4 $K/o4 --repeat 3 > $R/gpurun_out/r6l/m4_stdout.txt 2>&1
python3 $R/tools/prove_timeline.py /tmp/tl4 --gap-ms 0.4 --all > $R/gpurun_out/r6l/timeline_mnt4753.txt 2>&1
head -60 $R/gpurun_out/r6l/timeline_mnt6753.txt; grep "Total time" $R/gpurun_out/r6l/m6_stdout.txt; head -50 $R/gpurun_out/r6l/timeline_mnt4753.txt; grep "Total time" $R/gpurun_out/r6l/m4_stdout.txt
