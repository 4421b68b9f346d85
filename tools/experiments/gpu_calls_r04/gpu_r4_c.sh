#!/bin/bash
# round 4, GPU call C: T floor 8 + spill-free wide reduction steps (parity), SQ counters of the level kernels, window width of small G1 MSMs
# with the tree merge, the pieces of compute_H in the bench extras
mkdir -p gpurun_out/r4c
export TMPDIR=/tmp
O=gpurun_out/r4c
( time python -m pytest tests/test_msm_gpu.py tests/test_prover_gpu.py -m gpu -x -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.log
python - > $O/small_c.txt 2>&1 <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from __graft_entry__ import load_package
pkg = load_package(); pkg.init(0)
def run(curve, group, logn, env):
    for k in ("MNT753_MSM_TMIN", "MNT753_MSM_PRE_C"):
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    n = 1 << logn
    pts = pkg.synth_points(curve, group, 42, n); sc = pkg.synth_scalars(curve, 43, n)
    bs = pkg.BaseSet(curve, group, pts); d = pkg.DeviceBuffer.from_numpy(sc)
    best = None
    for rep in range(5):
        res = bs.msm(d.ptr.value, n=n, on_device=True); t = pkg.msm_last_timing()
        if rep and (best is None or t["total_ms"] < best["total_ms"]): best = t
    ok = bool(np.array_equal(pkg.point_to_affine(curve, group, res), pkg.point_to_affine(curve, group, pkg.synth_expected_msm(curve, group, 42, sc))))
    plan = pkg.msm_last_plan(); bs.close(); d.close()
    print(json.dumps({"curve": curve, "group": group, "log2_n": logn, "env": env, **{k: round(v, 3) for k, v in best.items()}, "c": plan["window_bits"], "T": plan["entries_per_lane"], "ok": ok}), flush=True)
for logn in (12, 13, 14, 15):
    for c in (None, 13, 14, 15, 16, 17):
        for tmin in (None, 4):
            env = {}
            if c: env["MNT753_MSM_PRE_C"] = c
            if tmin: env["MNT753_MSM_TMIN"] = tmin
            run(1, 1, logn, env)
PY
echo "small rc=$?"; cat $O/small_c.txt | cut -c1-260
mkdir -p build_exp/r4 && cp snark-challenge-prover-reference_amd/libmnt753_hip.so build_exp/r4/
sh tools/experiments/sq_ab.sh r4 > $O/sq_ab.log 2>&1; cp gpurun_out/sq_ab/r4.txt $O/sq_r4.txt 2>/dev/null; cat $O/sq_r4.txt | cut -c1-400
( time python bench.py --steps 10 --warmup 3 --no-prove --no-cpu-baseline --no-traffic --no-exchange ) > $O/bench_extras.json 2> $O/bench_extras.err; echo "bench rc=$?"
python -c "
import json
j=json.loads([l for l in open('$O/bench_extras.json') if l.startswith('{')][-1]); e=j['extras']
print(j['value'], j['ms_per_step'], j['phases_ms'])
print({k: e[k] for k in e if 'compute_h' in k or 'load' in k or 'fft' in k or 'g2' in k or 'no_table' in k})
print(e['slice_sweep_ms']); print(e['predicted_prove_s'])"
