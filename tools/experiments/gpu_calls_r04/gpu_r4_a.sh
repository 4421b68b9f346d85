#!/bin/bash
# round 4, GPU call A: the whole -m gpu suite, the default bench line, the small-MSM env sweep.  Run from the repo root on the GPU box.
mkdir -p gpurun_out/r4a
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/r4a/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4a/pytest_gpu.log
tail -30 gpurun_out/r4a/pytest_gpu.log
( time python bench.py ) > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err
echo "bench rc=$?"
tail -c 600 gpurun_out/r4a/bench.err
python - <<'PY'
import json
try:
    j = json.loads([l for l in open('gpurun_out/r4a/bench.json') if l.startswith('{')][-1])
    print({k: j[k] for k in ('value', 'ms_per_step', 'parity_ok', 'exchange_us') if k in j})
    print('prove', j.get('prove', {}).get('input_to_output_s_all'), 'mnt6', j.get('prove_mnt6753', {}).get('input_to_output_s_all'))
    print('cpu_prove', [(c.get('curve'), c.get('log2_d'), c.get('input_to_output_s'), c.get('threads'), c.get('matches_minted_hash'), c.get('same_bytes_as_gpu')) for c in j.get('cpu_prove', [])])
    print('exchange', j.get('exchange'))
except Exception as e:
    print('bench parse failed', e)
PY
( time python tools/experiments/small_msm_sweep.py quick ) > gpurun_out/r4a/small_sweep.txt 2>&1
echo "sweep rc=$?"
tail -5 gpurun_out/r4a/small_sweep.txt
