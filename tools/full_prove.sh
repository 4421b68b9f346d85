#!/bin/sh
# Full-size prove on the GPU box: synthetic parameter/input files in the reference's format, main_hip (GPU) vs the
# CPU oracle on all host cores, sha256 of both outputs.  usage: tools/full_prove.sh MNT4753 20 [skip-cpu]
set -e
cd "$(dirname "$0")/.."
CURVE=${1:-MNT4753}; LOG=${2:-20}; SKIPCPU=$3
D=${TMPDIR:-/tmp}/prove_${CURVE}_${LOG}; mkdir -p $D
python tools/synth_files.py $CURVE $LOG $D/params $D/input
ls -la $D
echo "== GPU (main_hip, fused compute_H)"; ./snark-challenge-prover-reference_amd/main_hip $CURVE compute $D/params $D/input $D/out_gpu --fused-h
echo "== GPU (main_hip, B:: call sequence of the reference driver)"; ./snark-challenge-prover-reference_amd/main_hip $CURVE compute $D/params $D/input $D/out_gpu2
if [ -z "$SKIPCPU" ]; then
  make -s -C oracle oracle_main
  echo "== CPU oracle ($(nproc) cores)"; ./oracle/oracle_main $CURVE compute $D/params $D/input $D/out_cpu
fi
sha256sum $D/out_*
rm -rf $D
