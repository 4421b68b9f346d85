# GPU box: per-kernel times of experimental builds (build_exp/<variant>/libmnt753_hip.so); results may be wrong by design
for v in "$@"; do RUN_TIMEOUT=150 sh tools/experiments/run_with_lib.sh $v sh tools/kstats.sh exp_$v > /dev/null 2>&1; echo "== $v rc=$?"; head -6 gpurun_out/kstats/exp_${v}_stats.csv | cut -d, -f1-4 | cut -c1-110; done
