#!/bin/sh
# Runs the REFERENCE provers (oracle/_ref/main = libsnark/main.cpp, bos_coster; oracle/_ref/piecewise_host =
# cuda_prover_piecewise.cu over the B:: wrapper, BDLO12) on the tiny generate_parameters sets in tests/golden/
# and records their outputs.  Run after oracle/_ref/mint_golden; needs /root/reference (container only).
set -e
cd "$(dirname "$0")/.."
G=tests/golden
for c in 4 6; do
  ./oracle/_ref/main MNT${c}753 compute $G/e2e_mnt${c}_params.bin $G/e2e_mnt${c}_input.bin $G/e2e_mnt${c}_output.bin > /dev/null
  ./oracle/_ref/piecewise_host MNT${c}753 compute $G/e2e_mnt${c}_params.bin $G/e2e_mnt${c}_input.bin /tmp/e2e_pw_${c}.bin > /dev/null 2>&1
  cmp $G/e2e_mnt${c}_output.bin /tmp/e2e_pw_${c}.bin
done
( cd $G && sha256sum *.bin > SHA256SUMS )
echo "e2e outputs minted; main == piecewise_host on both curves"
