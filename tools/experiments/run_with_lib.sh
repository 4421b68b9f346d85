# GPU box: run a command against build_exp/<variant>/libmnt753_hip.so instead of the product library.  The product file is never
# touched: the Python binding honours MNT753_LIB, main_hip / piecewise_hip (RUNPATH $ORIGIN) honour LD_LIBRARY_PATH.
#   sh tools/experiments/run_with_lib.sh <variant> <command...>
V=$1; shift
D=$PWD/build_exp/$V
[ -f "$D/libmnt753_hip.so" ] || { echo "run_with_lib: $D/libmnt753_hip.so not built (tools/experiments/build_variant.sh $V ...)" >&2; exit 2; }
MNT753_LIB=$D/libmnt753_hip.so LD_LIBRARY_PATH=$D${LD_LIBRARY_PATH:+:$LD_LIBRARY_PATH} timeout ${RUN_TIMEOUT:-200} "$@"
