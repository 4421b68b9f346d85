#!/bin/sh
# Full-size prove through prove_mgpu.py at world 1 and 2 (two ranks share the box's single GPU over gloo), sha256 vs main_hip
set -e
cd "$(dirname "$0")/.."
CURVE=${1:-MNT4753}; LOG=${2:-20}
D=${TMPDIR:-/tmp}/provem_${CURVE}_${LOG}; mkdir -p $D
python tools/synth_files.py $CURVE $LOG $D/params $D/input
./snark-challenge-prover-reference_amd/main_hip $CURVE compute $D/params $D/input $D/out_main_hip --fused-h --quiet
python prove_mgpu.py $CURVE compute $D/params $D/input $D/out_mgpu_1
PROVE_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 prove_mgpu.py $CURVE compute $D/params $D/input $D/out_mgpu_2 2>&1 | grep -v Gloo
sha256sum $D/out_*
rm -rf $D
