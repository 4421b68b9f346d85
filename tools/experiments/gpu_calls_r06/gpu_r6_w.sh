#!/bin/bash
# round 6, GPU call W: the tree with the reworked sort stage -- the whole -m gpu suite, smoke(), and the evidence bundle (bench line,
# rocprofv3 kernel stats of the bench command, PMC passes, G2 stats, full proves, one-shot wall): tools/collect_profiles.sh
mkdir -p gpurun_out/r6w; export TMPDIR=/tmp
O=$PWD/gpurun_out/r6w
( timeout 2400 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1
echo "pytest -m gpu rc=$?"; tail -3 $O/pytest_gpu.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
sh tools/collect_profiles.sh > $O/collect.log 2>&1; echo "collect rc=$?"
python3 - <<'PY'
import json
j = json.load(open("gpurun_out/prof/bench_line.json"))
print({k: j[k] for k in ("value", "ms_per_step", "parity_ok")}, j["roofline"]["modmul_frac"], j["prove"]["input_to_output_s_all"], j["prove"]["one_shot_wall_s"], j["prove_mnt6753"]["input_to_output_s_all"], j["extras"]["g2_msm_2p20_ms"])
PY
ls gpurun_out/prof | head -50
