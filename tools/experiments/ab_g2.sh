# GPU box: A/B of library variants on the G2 MSMs only (MNT4753 2^20, MNT6753 2^15), two rounds, alternating: sh tools/experiments/ab_g2.sh <variant> ...
for round in 1 2; do for v in "$@"; do
  a=$(CURVE=0 GROUP=2 sh tools/experiments/run_with_lib.sh $v python3 tools/dev_msm_big.py 20 4 2>/dev/null | tail -1 | sed 's/.*total_ms=//')
  b=$(CURVE=1 GROUP=2 sh tools/experiments/run_with_lib.sh $v python3 tools/dev_msm_big.py 15 4 2>/dev/null | tail -1 | sed 's/.*total_ms=//')
  echo "round $round $v: MNT4 G2 2^20 $a | MNT6 G2 2^15 $b"
done; done
