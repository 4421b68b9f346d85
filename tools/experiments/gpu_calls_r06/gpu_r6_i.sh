#!/bin/bash
# round 6, GPU call I: the final tree once more -- the whole -m gpu suite, __graft_entry__.smoke(), and the driver's bench command
mkdir -p gpurun_out/r6i
export TMPDIR=/tmp
O=$PWD/gpurun_out/r6i
( timeout 2400 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?"; tail -3 $O/pytest_gpu.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_stderr.log; echo "bench rc=$?"
python3 - <<'PY'
import json
j = json.load(open("gpurun_out/r6i/bench_line.json"))
print({k: j[k] for k in ("value", "ms_per_step", "parity_ok")}, j["roofline"]["modmul_frac"], j["prove"]["input_to_output_s_all"], j["prove"]["one_shot_wall_s"], j["prove_mnt6753"]["input_to_output_s_all"], j["extras"]["g2_msm_2p20_ms"])
PY
