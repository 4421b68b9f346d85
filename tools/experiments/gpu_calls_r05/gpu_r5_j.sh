#!/bin/bash
# round 5, GPU call J (final code): the whole GPU suite (untraced), then the evidence bundle (tools/collect_profiles.sh)
mkdir -p gpurun_out/r5j
export TMPDIR=/tmp
O=$PWD/gpurun_out/r5j
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log | cut -c1-200
sleep 20
sh tools/collect_profiles.sh > $O/collect.log 2>&1; echo "collect rc=$?"
python3 -c "
import json; j=json.load(open('gpurun_out/prof/bench_line.json')); print('BENCH', round(j['ms_per_step'],3), round(j['value']/1e6,2), 'prove', j['prove'].get('input_to_output_s'), j['prove'].get('load_params_s'), j['prove'].get('cold_process'), 'mnt6', j['prove_mnt6753'].get('input_to_output_s'), 'parity', j['parity_ok'])"
