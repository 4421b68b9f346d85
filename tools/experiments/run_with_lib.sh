# GPU box: run a command with build_exp/lib_<variant>.so in place of the product library (the box's copy is scratch)
#   sh tools/experiments/run_with_lib.sh <variant> <command...>
P=snark-challenge-prover-reference_amd; V=$1; shift
cp $P/libmnt753_hip.so /tmp/lib_orig.so; cp build_exp/lib_$V.so $P/libmnt753_hip.so
timeout 200 "$@"; rc=$?
cp /tmp/lib_orig.so $P/libmnt753_hip.so; exit $rc
