#!/bin/bash
# round 6, GPU call N: A's MSM enqueued ahead of C's (--c-last) against the default, alternating in one process each, resident proofs
mkdir -p gpurun_out/r6n; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
K=/tmp/pk; mkdir -p $K
python3 tools/synth_files.py MNT6753 15 $K/p6 $K/i6 > /dev/null
python3 tools/synth_files.py MNT4753 20 $K/p4 $K/i4 > /dev/null
M=$R/snark-challenge-prover-reference_amd/main_hip
{
for round in 1 2 3; do
  for v in "" "--c-last"; do
    echo "== MNT4753 2^20 round $round flags '$v'"; $M MNT4753 compute $K/p4 $K/i4 $K/o4 --repeat 5 $v | grep "Total time from" | tr '\n' ' '; echo; sha256sum $K/o4 | cut -c1-16
  done
done
for round in 1 2 3; do
  for v in "" "--c-last"; do
    echo "== MNT6753 2^15 round $round flags '$v'"; $M MNT6753 compute $K/p6 $K/i6 $K/o6 --repeat 8 $v | grep "Total time from" | tr '\n' ' '; echo; sha256sum $K/o6 | cut -c1-16
  done
done
} > gpurun_out/r6n/c_last_ab.txt 2>&1
cat gpurun_out/r6n/c_last_ab.txt
