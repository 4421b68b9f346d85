# GPU box: per-kernel times of experimental builds of the G1 instantiation (build_exp/lib_<variant>.so); results are wrong by design
P=snark-challenge-prover-reference_amd
cp $P/libmnt753_hip.so /tmp/lib_orig.so
for v in "$@"; do cp build_exp/lib_$v.so $P/libmnt753_hip.so; timeout 150 sh tools/kstats.sh exp_$v > /dev/null 2>&1; echo "== $v rc=$?"; head -6 gpurun_out/kstats/exp_${v}_stats.csv | cut -d, -f1-4 | cut -c1-110; done
cp /tmp/lib_orig.so $P/libmnt753_hip.so
