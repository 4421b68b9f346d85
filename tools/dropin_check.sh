#!/bin/sh
# Proves the drop-in claim of include/prover_hip_functions.hpp (container only; needs /root/reference):
# the reference's OWN driver -- cuda_prover_piecewise.cu lines 14-120, i.e. compute_H<B>, run_prover<B> and main, byte for
# byte -- is compiled against include/prover_hip_functions.hpp and linked with libmnt753_hip.so.  The only edit is the one
# INTEGRATION.md section 1 tells a maintainer to make: the two `run_prover<mnt{4,6}753_libsnark>` instantiations become
# `run_prover<mnt{4,6}753_hip>`, and the include names our header.  Output goes to oracle/_ref/ (git-ignored; the reference's
# text is never stored in the repository).  The binary is the reference's CLI over the MI355X library: on the GPU box it
# writes the same proofs as main_hip (tests/test_prover_gpu.py::test_reference_driver_unchanged runs it when present).
set -e
R=${1:-/root/reference}
HERE=$(cd "$(dirname "$0")/.." && pwd)
O=$HERE/oracle/_ref
[ -f "$R/cuda_prover_piecewise.cu" ] || { echo "dropin_check: $R absent"; exit 0; }
mkdir -p $O
GEN=$O/piecewise_hip.gen.cpp
{ echo '#include <string>'; echo '#include <prover_hip_functions.hpp>';
  sed -n '14,120p' $R/cuda_prover_piecewise.cu | sed -e 's/run_prover<mnt4753_libsnark>/run_prover<mnt4753_hip>/' -e 's/run_prover<mnt6753_libsnark>/run_prover<mnt6753_hip>/'; } > $GEN
# exactly two lines may differ from the reference's text
NDIFF=$(sed -n '14,120p' $R/cuda_prover_piecewise.cu | diff - $GEN | grep -c '^>' || true)
[ "$NDIFF" = "4" ] || { echo "dropin_check: expected 2 include lines + 2 swapped instantiations, found $NDIFF differing lines"; exit 1; }
PKG=$HERE/snark-challenge-prover-reference_amd
g++ -O2 -std=c++17 -pthread -I$HERE/include $GEN $PKG/host/prover_hip_functions.cpp -L$PKG -lmnt753_hip -Wl,-rpath,'$ORIGIN/../../snark-challenge-prover-reference_amd' -o $O/piecewise_hip
rm -f $GEN   # the generated translation unit holds the reference's text: it is not kept, not even in the git-ignored directory
echo "dropin_check: the reference driver compiled unchanged against prover_hip_functions.hpp -> oracle/_ref/piecewise_hip"
