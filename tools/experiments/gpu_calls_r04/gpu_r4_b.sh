#!/bin/bash
# round 4, GPU call B: tree edge merge (parity + timing), CU budget for the small curve's prove, MFMA reduction microbenchmark, plan-B sizing
mkdir -p gpurun_out/r4b
export TMPDIR=/tmp
O=gpurun_out/r4b
( time python -m pytest tests/test_msm_gpu.py tests/test_prover_gpu.py tests/test_device_kat_gpu.py -m gpu -x -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.log
./build/mul_mfma > $O/mul_mfma.txt 2>&1; echo "mul_mfma rc=$?"; cat $O/mul_mfma.txt
python - > $O/small.txt 2>&1 <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from __graft_entry__ import load_package
pkg = load_package(); pkg.init(0)
def run(curve, group, logn, env):
    for k in ("MNT753_MSM_TMIN", "MNT753_EDGE_TREE"):
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
# EDGE_TREE is read once per process (static): the old merge is measured in a second process below
for curve, sizes in ((1, (12, 13, 14, 15)), (0, (14, 17, 20))):
    for logn in sizes:
        for group in (1, 2):
            for tmin in (None, 8, 4):
                if tmin and logn > 14: continue
                run(curve, group, logn, {"MNT753_MSM_TMIN": tmin} if tmin else {})
PY
echo "small rc=$?"; tail -3 $O/small.txt
MNT753_EDGE_TREE=0 python - > $O/small_oldmerge.txt 2>&1 <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from __graft_entry__ import load_package
pkg = load_package(); pkg.init(0)
for curve, group, logn in ((1, 1, 12), (1, 2, 12), (1, 1, 15), (1, 2, 15), (0, 1, 20), (0, 2, 20)):
    n = 1 << logn
    pts = pkg.synth_points(curve, group, 42, n); sc = pkg.synth_scalars(curve, 43, n)
    bs = pkg.BaseSet(curve, group, pts); d = pkg.DeviceBuffer.from_numpy(sc)
    best = None
    for rep in range(5):
        res = bs.msm(d.ptr.value, n=n, on_device=True); t = pkg.msm_last_timing()
        if rep and (best is None or t["total_ms"] < best["total_ms"]): best = t
    bs.close(); d.close()
    print(json.dumps({"curve": curve, "group": group, "log2_n": logn, "old_merge": True, **{k: round(v, 3) for k, v in best.items()}}), flush=True)
PY
cat $O/small_oldmerge.txt
# the small curve's prove with a CU budget for the point kernels (the latency-bound chains of the other MSMs run beside them)
python tools/synth_files.py MNT6753 15 /tmp/p6 /tmp/i6 > /dev/null
for cus in 256 240 224 192 160 128; do
  echo "== point-cus $cus" >> $O/prove6_cus.txt
  ./snark-challenge-prover-reference_amd/main_hip MNT6753 compute /tmp/p6 /tmp/i6 /tmp/o6 --repeat 4 --point-cus $cus 2>&1 | grep "Total time from input" >> $O/prove6_cus.txt
  sha256sum /tmp/o6 | cut -c1-16 >> $O/prove6_cus.txt
done
cat $O/prove6_cus.txt
( time python tools/experiments/plan_b_sizing.py ) > $O/plan_b.txt 2>&1; echo "plan_b rc=$?"; cat $O/plan_b.txt | cut -c1-400
( time python bench.py --steps 10 --warmup 3 --no-prove --no-cpu-baseline --no-traffic --no-exchange --no-extras ) > $O/bench_quick.json 2> $O/bench_quick.err; echo "bench rc=$?"
python -c "
import json
j=json.loads([l for l in open('$O/bench_quick.json') if l.startswith('{')][-1]); print(j['value'], j['ms_per_step'], j['phases_ms'])"
