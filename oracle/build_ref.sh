#!/bin/sh
# Builds the REFERENCE's own CPU prover from its sources where they lie (default /root/reference) into
# oracle/_ref/ -- never copies sources into the repo.  Used only to pin the oracle (oracle/mnt753_oracle.c)
# and, optionally, as the cpu_baseline of bench.py (kind "reference").
#
# The reference's CMake cannot configure here (enable_language(CUDA), pkg-config, Boost), so the handful of
# translation units on the prover path are compiled directly with g++ using the flags build.sh/CMakeLists.txt
# select (MULTICORE=ON, USE_PT_COMPRESSION=OFF, BINARY_OUTPUT, MONTGOMERY_OUTPUT, USE_ASM).  GMP: the image's
# libgmp.so.10 with the matching gmp.h from /opt/conda/include (a header the image ships, copied into _ref/inc
# so that no other conda header leaks into the include path).
set -e
R=${1:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
O=$HERE/_ref
GMP_SO=/usr/lib/x86_64-linux-gnu/libgmp.so.10
[ -d "$R/libsnark" ] || { echo "build_ref: $R not found"; exit 0; }
[ -f /opt/conda/include/gmp.h ] && [ -f $GMP_SO ] || { echo "build_ref: GMP header/library missing -- reference unbuildable here"; exit 0; }
if [ -x $O/main ] && [ -x $O/generate_parameters ] && [ -x $O/piecewise_host ] && [ -x $O/mint_golden ] && [ $O/mint_golden -nt $HERE/mint_golden.cpp ] \
   && [ -x $O/ref_msm_bench ] && [ $O/ref_msm_bench -nt $HERE/ref_msm_bench.cpp ] \
   && [ -x $O/ref_groth16 ] && [ $O/ref_groth16 -nt $HERE/ref_groth16.cpp ]; then
  echo "build_ref: oracle/_ref up to date"; exit 0
fi
mkdir -p $O/inc $O/obj
cp /opt/conda/include/gmp.h /opt/conda/include/gmpxx.h $O/inc/
F="-std=c++14 -O2 -fopenmp -DMULTICORE=1 -DBINARY_OUTPUT -DMONTGOMERY_OUTPUT -DNO_PT_COMPRESSION=1 -DUSE_ASM -DNO_PROCPS -DCURVE_MNT4 -I$O/inc -I$R -I$R/depends/libff -I$R/depends/libfqfft -w"
L=$R/depends/libff/libff
SRCS="algebra/curves/mnt753/mnt4753/mnt4753_g1.cpp algebra/curves/mnt753/mnt4753/mnt4753_g2.cpp algebra/curves/mnt753/mnt4753/mnt4753_init.cpp algebra/curves/mnt753/mnt4753/mnt4753_pairing.cpp algebra/curves/mnt753/mnt4753/mnt4753_pp.cpp algebra/curves/mnt753/mnt46753_common.cpp algebra/curves/mnt753/mnt6753/mnt6753_g1.cpp algebra/curves/mnt753/mnt6753/mnt6753_g2.cpp algebra/curves/mnt753/mnt6753/mnt6753_init.cpp algebra/curves/mnt753/mnt6753/mnt6753_pairing.cpp algebra/curves/mnt753/mnt6753/mnt6753_pp.cpp algebra/curves/mnt/mnt4/mnt4_g1.cpp algebra/curves/mnt/mnt4/mnt4_g2.cpp algebra/curves/mnt/mnt4/mnt4_init.cpp algebra/curves/mnt/mnt4/mnt4_pairing.cpp algebra/curves/mnt/mnt4/mnt4_pp.cpp algebra/curves/mnt/mnt46_common.cpp common/double.cpp common/profiling.cpp common/utils.cpp"
# every background compile is waited for by PID so that a failure stops the script (a bare `wait` returns 0)
PIDS=""
wait_all() { for p in $PIDS; do wait $p || { echo "build_ref: a compile job failed"; exit 1; }; done; PIDS=""; }
for s in $SRCS; do
  o=$O/obj/$(echo $s | tr / _).o
  [ -f $o ] || { g++ $F -c $L/$s -o $o & PIDS="$PIDS $!"; }
done
wait_all
g++ $F $R/libsnark/main.cpp $O/obj/*.o -o $O/main $GMP_SO & PIDS="$PIDS $!"
g++ $F $R/libsnark/generate_parameters.cpp $O/obj/*.o -o $O/generate_parameters $GMP_SO & PIDS="$PIDS $!"
g++ $F -c $R/libsnark/prover_reference_functions.cpp -o $O/prf.o & PIDS="$PIDS $!"
wait_all
# the wrapper-based driver: lines 14-120 of cuda_prover_piecewise.cu are plain host C++ (no kernels)
{ echo '#include <string>'; echo '#include <prover_reference_functions.hpp>'; sed -n '14,120p' $R/cuda_prover_piecewise.cu; } > $O/piecewise_host.gen.cpp
g++ $F -I$R/libsnark/prover_reference_include $O/piecewise_host.gen.cpp $O/prf.o $O/obj/*.o -o $O/piecewise_host $GMP_SO & PIDS="$PIDS $!"
# golden-vector minting tool: OUR program, linked against the reference's libff/libfqfft/libsnark code
g++ $F -I$R/libsnark/prover_reference_include $HERE/mint_golden.cpp $O/obj/*.o -o $O/mint_golden $GMP_SO & PIDS="$PIDS $!"
# CPU-baseline driver for bench.py: OUR program around the reference's multi_exp (BDLO12), as B::multiexp_G1 calls it
g++ $F $HERE/ref_msm_bench.cpp $O/obj/*.o -o $O/ref_msm_bench $GMP_SO & PIDS="$PIDS $!"
# fixtures + checker for the steps either side of the hot path (witness map, full proof + verifier): OUR program on the reference's libsnark
g++ $F $HERE/ref_groth16.cpp $O/obj/*.o -o $O/ref_groth16 $GMP_SO & PIDS="$PIDS $!"
wait_all
rm -f $O/piecewise_host.gen.cpp   # holds the reference's text: not kept once it is compiled
echo "build_ref: built $(ls $O | tr '\n' ' ')"
