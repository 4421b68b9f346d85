# Build container: an experimental variant of the library, build_exp/<variant>/libmnt753_hip.so.
#   sh tools/experiments/build_variant.sh <variant> "<extra hipcc flags, e.g. -DMNT753_PAIR_TIMING>" [source.hip ...]
# The named translation units (default: the G1 and G2 instantiations of MNT4753) are recompiled with the extra flags, every other
# object comes from build/ (run `make` first).
set -e
V=$1; FLAGS=$2; shift 2 || true
SRCS=${*:-"msm_inst_mnt4g1.hip msm_inst_mnt4g2.hip"}
P=snark-challenge-prover-reference_amd; D=build_exp/$V
mkdir -p $D
OBJS=""
for o in build/*.o; do
  b=$(basename $o .o)
  case " $SRCS " in *" $b.hip "*) ;; *) case $b in *_t|mnt753_testhooks|mnt753_synth_points) ;; *) OBJS="$OBJS $o";; esac;; esac
done
for s in $SRCS; do
  b=$(basename $s .hip)
  hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-result $FLAGS -c $P/csrc/$s -o $D/$b.o &
done
wait
for s in $SRCS; do OBJS="$OBJS $D/$(basename $s .hip).o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-soname,libmnt753_hip.so -o $D/libmnt753_hip.so $OBJS -ldl
echo "built $D/libmnt753_hip.so"
