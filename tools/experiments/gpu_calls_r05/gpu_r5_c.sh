#!/bin/bash
# round 5, GPU call C (final code): (1) parameter load of the FIRST prover on a fresh box, of a second right behind it and of a third
# after a pause (where do the ~3 s go that appear when another process has just returned ~100 GB of device memory?); (2) the whole GPU
# suite under rocprofv3 --kernel-trace, one trace per process (children: main_hip, bench.py) -> tools/kernel_coverage.py; the bench tests
# without the profiler; (3) the reference generator's FULL-size parameter sets through the reference's ./main and main_hip
# (MNT753_REAL_PARAMS=1); (4) the evidence bundle (tools/collect_profiles.sh).
mkdir -p gpurun_out/r5c
export TMPDIR=/tmp
O=$PWD/gpurun_out/r5c
R=$PWD
K=/tmp/prove_keep; mkdir -p $K
python3 tools/synth_files.py MNT4753 20 $K/p4 $K/i4 > /dev/null
M=$R/snark-challenge-prover-reference_amd/main_hip
{ echo "== first prover of the box"; MNT753_TRACE=1 $M MNT4753 compute $K/p4 $K/i4 $K/o4 2>&1 | grep -i "load params\|Total time";
  echo "== a second one right behind it"; MNT753_TRACE=1 $M MNT4753 compute $K/p4 $K/i4 $K/o4 2>&1 | grep -i "load params\|Total time";
  sleep 20; echo "== a third after 20 s"; MNT753_TRACE=1 $M MNT4753 compute $K/p4 $K/i4 $K/o4 2>&1 | grep -i "load params\|Total time";
  echo "== cold process (MNT753_NO_WARMUP=1), two proofs"; MNT753_NO_WARMUP=1 $M MNT4753 compute $K/p4 $K/i4 $K/o4 --repeat 2 2>&1 | grep -i "load params:\|Total time"; sha256sum $K/o4; } > $O/load_params_fresh_box.log 2>&1
cat $O/load_params_fresh_box.log
rm -rf $K
cd /tmp
( time timeout 2400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kcov -o k_%pid% -- python3 -m pytest $R/tests -m gpu -q -p no:cacheprovider -k "not test_bench" ) > $O/pytest_traced.log 2>&1
echo "traced pytest rc=$?"; tail -4 $O/pytest_traced.log | cut -c1-200; ls /tmp/kcov | wc -l
cd $R
python3 tools/kernel_coverage.py --traces /tmp/kcov --out $O/kernel_coverage.txt > /dev/null 2>$O/kernel_coverage.err; grep -A12 "NEVER launched" $O/kernel_coverage.txt | head -20
( time timeout 1500 python -m pytest tests/test_bench_gpu.py -m gpu -q -x ) > $O/pytest_bench.log 2>&1
echo "bench tests rc=$?"; tail -3 $O/pytest_bench.log | cut -c1-200
( time MNT753_REAL_PARAMS=1 MNT753_REAL_PARAMS_REPORT=$O/real_params_report.json timeout 2400 python -m pytest tests/test_live_reference_gpu.py -m gpu -q -x -s ) > $O/pytest_real_params.log 2>&1
echo "real params rc=$?"; tail -5 $O/pytest_real_params.log | cut -c1-300; cat $O/real_params_report.json 2>/dev/null
sh tools/collect_profiles.sh > $O/collect.log 2>&1; echo "collect rc=$?"
python3 -c "
import json; j=json.load(open('gpurun_out/prof/bench_line.json')); print('BENCH', round(j['ms_per_step'],3), round(j['value']/1e6,2), 'prove', j['prove'].get('input_to_output_s'), j['prove'].get('load_params_s'), j['prove'].get('cold_process'), 'mnt6', j['prove_mnt6753'].get('input_to_output_s'), 'parity', j['parity_ok'])"
