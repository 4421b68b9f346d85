#!/bin/bash
# round 6, GPU call G: the buffers of the batched-affine levels -- one set per base set (MNT753_PAIR_POOL=0, round 5), one per device
# (=1), or one per device with the small sets keeping their own (=2, default): the resident prove (4 proofs) and the footprint, three
# rounds alternating on one box; MNT6753 the same
mkdir -p gpurun_out/r6g
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6g
D=/tmp/fp; mkdir -p $D
python3 tools/synth_files.py MNT4753 20 $D/params $D/input > /dev/null 2>&1
python3 tools/synth_files.py MNT6753 15 $D/p6 $D/i6 > /dev/null 2>&1
M=./snark-challenge-prover-reference_amd/main_hip
{
for round in 1 2 3; do for mode in 0 1 2; do
  sleep 10
  echo "== round $round MNT753_PAIR_POOL=$mode"
  MNT753_PAIR_POOL=$mode MNT753_TRACE_LOAD=1 $M MNT4753 compute $D/params $D/input $D/out --repeat 5 2>&1 | grep -E "^load params|Total time|device memory" | tr '\n' ' '; echo; sha256sum $D/out | cut -c1-16
  MNT753_PAIR_POOL=$mode $M MNT6753 compute $D/p6 $D/i6 $D/o6 --repeat 6 2>&1 | grep -E "Total time" | tr '\n' ' '; echo
done; done
} > $O/level_buffers_pooling_ab.txt 2>&1
cat $O/level_buffers_pooling_ab.txt
rm -rf $D
( timeout 900 python -m pytest tests/test_msm_gpu.py tests/test_prover_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -2 $O/pytest.log
( MNT753_PAIR_POOL=1 timeout 900 python -m pytest tests/test_msm_gpu.py -m gpu -q -x ) > $O/pytest_pool1.log 2>&1
echo "pytest (MNT753_PAIR_POOL=1) rc=$?"; tail -2 $O/pytest_pool1.log
