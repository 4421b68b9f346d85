#!/bin/bash
# round 6, GPU call Y: prover and bench tests, the bench line and three proofs with the spinning wait as the default
mkdir -p gpurun_out/r6y; export TMPDIR=/tmp; O=$PWD/gpurun_out/r6y
( timeout 1500 python -m pytest tests/test_prover_gpu.py tests/test_bench_gpu.py tests/test_selftest_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log | cut -c1-200
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_stderr.log; echo "bench rc=$?"
python3 - <<'PY'
import json
j = json.load(open("gpurun_out/r6y/bench_line.json")); p = j["phases_ms"]
print({k: j[k] for k in ("value", "ms_per_step", "parity_ok")}, {k: round(v, 3) for k, v in p.items()}, j["roofline"]["modmul_frac"], j["prove"]["input_to_output_s_all"], j["prove"]["one_shot_wall_s"], j["prove_mnt6753"]["input_to_output_s_all"], j["extras"]["g2_msm_2p20_ms"])
PY
uptime
