# GPU box: board power and clocks (rocm-smi, 5 samples per second) while (a) the multiplier microbenchmark, (b) the 20-step G1 MSM loop
# of bench.py and (c) three full proves run -- is the chip power-limited under the 753-bit multiply-add load?
#   sh tools/experiments/power_sample.sh > gpurun_out/power.txt
sample() { while :; do rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|GPU use" | tr '\n' ' '; echo; sleep 0.2; done; }
rocm-smi --showmaxpower --showclkfrq 2>/dev/null | grep -E "Max|sclk|\*" | head -20
echo "== idle"; (sample & S=$!; sleep 1; kill $S) | tail -2
echo "== build/mul_var (pure Montgomery products, 1 to 8 waves per SIMD)"; (sample & S=$!; timeout 120 ./build/mul_var > /dev/null 2>&1; kill $S) | awk 'NR % 3 == 0' | tail -12
echo "== bench.py G1 MSM loop (60 steps)"; (sample & S=$!; timeout 200 python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-prove --no-extras --no-traffic > /dev/null 2>&1; kill $S) | tail -12
