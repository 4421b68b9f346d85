# GPU box: A/B of library variants in ONE call (box-to-box variance is several percent): sh tools/experiments/ab.sh <variant> <variant> ...
# per variant and round: G1 2^20 MSM phases (bench.py) and the G2 2^20 MSM
for round in 1 2; do for v in "$@"; do
  g1=$(sh tools/experiments/run_with_lib.sh $v python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prove --no-extras --no-traffic 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],2), {k: round(v,2) for k,v in j['phases_ms'].items() if k in ('sort_ms','accumulate_ms','reduce_ms')})")
  g2=$(CURVE=0 GROUP=2 sh tools/experiments/run_with_lib.sh $v python3 tools/dev_msm_big.py 20 4 2>/dev/null | tail -1 | sed 's/.*total_ms=//')
  echo "round $round $v: G1 $g1 | G2 $g2"
done; done
