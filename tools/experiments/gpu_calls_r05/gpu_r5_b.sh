#!/bin/bash
# round 5, GPU call B: after the pruning of the measured-and-rejected variants (33 -> 12 environment switches, the one-lane G2 kernels,
# the pointer-jumping merge, the radix sort stage, Karatsuba, ...): (1) the whole GPU suite under rocprofv3 --kernel-trace, which is
# also the kernel-instantiation coverage list (tools/kernel_coverage.py); the bench tests without the profiler; (2) the headline on this
# box; (3) whole-row LDS-DMA pieces in the first level against the packed pieces, alternating; (4) where a parameter load goes.
mkdir -p gpurun_out/r5b
export TMPDIR=/tmp
O=$PWD/gpurun_out/r5b
R=$PWD
cd /tmp
( time timeout 2000 rocprofv3 --kernel-trace --output-format csv -d /tmp/kcov -o k -- python3 -m pytest $R/tests -m gpu -q -p no:cacheprovider --deselect $R/tests/test_bench_gpu.py ) > $O/pytest_traced.log 2>&1
echo "traced pytest rc=$?"; tail -4 $O/pytest_traced.log | cut -c1-200
cd $R
python3 tools/kernel_coverage.py --traces /tmp/kcov --out $O/kernel_coverage.txt > /dev/null 2>$O/kernel_coverage.err; grep -c . $O/kernel_coverage.txt; grep -A40 "NEVER launched" $O/kernel_coverage.txt | head -60
( time timeout 1500 python -m pytest tests/test_bench_gpu.py -m gpu -q -x ) > $O/pytest_bench.log 2>&1
echo "bench tests rc=$?"; tail -3 $O/pytest_bench.log | cut -c1-200
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prove --no-extras --no-traffic --no-exchange > $O/bench_quick.json 2> $O/bench_quick.err
python3 -c "
import json; j=json.load(open('$O/bench_quick.json')); print('bench', round(j['ms_per_step'],3), 'ms/step', round(j['value']/1e6,2), 'Mpts/s precompute_ms', round(j['precompute_ms'],1), j['phases_ms'], 'parity', j['parity_ok'])"
for round in 1 2 3; do for v in base wholerows; do
  if [ $v = base ]; then L=$R/snark-challenge-prover-reference_amd/libmnt753_hip.so; else L=$R/build_exp/$v/libmnt753_hip.so; fi
  (cd /tmp && MNT753_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wr_${v}_$round -o x -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prove --no-extras --no-traffic --no-exchange > /tmp/wr_${v}_$round.json 2>/dev/null)
  python3 - /tmp/wr_${v}_$round $v $round /tmp/wr_${v}_$round.json <<'PY'
import csv, glob, sys, json
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "k_pair_level" in r["Name"] and "true, false, false" in r["Name"]]
try:
    j = json.load(open(sys.argv[4])); extra = f"ms_per_step {j['ms_per_step']:.3f} parity {j['parity_ok']}"
except Exception as ex:
    extra = "bench line: " + repr(ex)[:80]
for r in rows[:1]:
    print(f"{sys.argv[2]:10s} round {sys.argv[3]}  level 1 avg_ms {float(r['AverageNs'])/1e6:8.3f} min_ms {float(r['MinNs'])/1e6:8.3f}   {extra}")
PY
done; done > $O/level1_whole_row_pieces.txt 2>&1
cat $O/level1_whole_row_pieces.txt
K=/tmp/prove_keep; mkdir -p $K
python3 tools/synth_files.py MNT4753 20 $K/p4 $K/i4 > /dev/null
M=$R/snark-challenge-prover-reference_amd/main_hip
MNT753_TRACE=1 $M MNT4753 compute $K/p4 $K/i4 $K/o4 --repeat 2 > $O/prove_trace.log 2>&1; grep -i "load params\|Total time" $O/prove_trace.log; sha256sum $K/o4
MNT753_TRACE=1 $M MNT4753 compute $K/p4 $K/i4 $K/o4 > $O/prove_trace2.log 2>&1; grep -i "load params" $O/prove_trace2.log
rm -rf $K
