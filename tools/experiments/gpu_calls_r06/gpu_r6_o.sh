#!/bin/bash
# round 6, GPU call O: --a-first (A's MSM, compute_H with the device to itself, then G2 and C) against the default order, alternating
mkdir -p gpurun_out/r6o; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
K=/tmp/pk; mkdir -p $K
python3 tools/synth_files.py MNT6753 15 $K/p6 $K/i6 > /dev/null
python3 tools/synth_files.py MNT4753 20 $K/p4 $K/i4 > /dev/null
M=$R/snark-challenge-prover-reference_amd/main_hip
{
for round in 1 2 3; do
  for v in "" "--a-first"; do
    echo "== MNT4753 2^20 round $round flags '$v'"; $M MNT4753 compute $K/p4 $K/i4 $K/o4 --repeat 5 $v | grep "Total time from" | tr '\n' ' '; echo; sha256sum $K/o4 | cut -c1-16
  done
done
for round in 1 2 3; do
  for v in "" "--a-first"; do
    echo "== MNT6753 2^15 round $round flags '$v'"; $M MNT6753 compute $K/p6 $K/i6 $K/o6 --repeat 8 $v | grep "Total time from" | tr '\n' ' '; echo; sha256sum $K/o6 | cut -c1-16
  done
done
} > gpurun_out/r6o/a_first_ab.txt 2>&1
cat gpurun_out/r6o/a_first_ab.txt
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl4 -o tl -- $M MNT4753 compute $K/p4 $K/i4 $K/o4 --repeat 3 --a-first > $R/gpurun_out/r6o/m4_stdout.txt 2>&1
python3 $R/tools/prove_timeline.py /tmp/tl4 --gap-ms 0.4 --all > $R/gpurun_out/r6o/timeline_mnt4753_a_first.txt 2>&1
head -12 $R/gpurun_out/r6o/timeline_mnt4753_a_first.txt
