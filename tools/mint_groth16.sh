#!/bin/sh
# Mints tests/golden/g16_mnt{4,6}/: a small circuit per curve from the REFERENCE's generator (oracle/_ref/ref_groth16 mint =
# generate_r1cs_example_with_field_input + r1cs_gg_ppzksnark_generator), the challenge proof the reference prover writes for it
# (oracle/_ref/main), a fixed s, the reference's completion of that proof (main.cpp:312-319) and its verifier's verdict.
# Container only (needs oracle/_ref, i.e. /root/reference); inputs are random, so the files are captured once and committed.
set -e
cd "$(dirname "$0")/.."
for c in 4 6; do
  D=tests/golden/g16_mnt$c; mkdir -p $D
  ./oracle/_ref/ref_groth16 mint MNT${c}753 5 $D > /dev/null
  ./oracle/_ref/main MNT${c}753 compute $D/params.bin $D/input.bin $D/challenge.bin > /dev/null 2>&1
  python3 - $c $D <<'PY'
import sys
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
pkg.synth_scalars(0 if sys.argv[1] == "4" else 1, 0x73, 1).tofile(sys.argv[2] + "/s.bin")
PY
  ./oracle/_ref/ref_groth16 complete MNT${c}753 $D $D/challenge.bin $D/s.bin $D/full.bin
  ./oracle/_ref/ref_groth16 verify MNT${c}753 $D $D/full.bin | tail -1
done
( cd tests/golden && sha256sum *.bin g16_mnt*/* > SHA256SUMS )
ls -la tests/golden/g16_mnt4 tests/golden/g16_mnt6
