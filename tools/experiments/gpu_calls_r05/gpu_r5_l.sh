#!/bin/bash
# round 5, GPU call L: the driver's command on the final bench.py (the prove children behind a pause) -> profiles/r05/bench_line.json
mkdir -p gpurun_out/r5l
sleep 20
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r5l/bench_line.json 2> gpurun_out/r5l/bench_stderr.log
python3 -c "
import json; j=json.loads([l for l in open('gpurun_out/r5l/bench_line.json') if l.startswith('{')][-1]); print('BENCH', round(j['ms_per_step'],3), round(j['value']/1e6,2), 'frac', j['roofline']['frac'], j['roofline']['modmul_frac'], 'prove', j['prove'].get('input_to_output_s_all'), j['prove'].get('load_params_s'), 'cold', j['prove'].get('cold_process',{}).get('input_to_output_s_all'), j['prove'].get('cold_process',{}).get('load_params_s'), 'mnt6', j['prove_mnt6753'].get('input_to_output_s_all'), 'parity', j['parity_ok'], 'rccl', j['exchange'].get('rccl_version'), 'g2', j['extras'].get('g2_msm_2p20_ms'), 'notable', j['no_window_table'])"
tail -3 gpurun_out/r5l/bench_stderr.log
