#!/bin/bash
# round 4, GPU call P: list-driven VM levels: parity (every form of the merge), timings
mkdir -p gpurun_out/r4p
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4p
R=$PWD
( time python -m pytest tests/test_device_kat_gpu.py tests/test_msm_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^FAILED|passed|failed" $O/pytest.log | cut -c1-150 | head -20
cat > /tmp/flow.py <<'PY'
import json, os, sys
sys.path.insert(0, os.environ["REPO"])
import numpy as np
from __graft_entry__ import load_package
pkg = load_package(); pkg.init(0)
tag = os.environ.get("TAG", "default")
cases = [(1, 1, 12), (1, 2, 12), (1, 1, 15), (1, 2, 15), (0, 1, 15), (0, 2, 15), (0, 2, 17)]
if os.environ.get("BIG"): cases += [(0, 1, 20), (0, 2, 20)]
for curve, group, logn in cases:
    n = 1 << logn
    pts = pkg.synth_points(curve, group, 42, n); sc = pkg.synth_scalars(curve, 43, n)
    bs = pkg.BaseSet(curve, group, pts); d = pkg.DeviceBuffer.from_numpy(sc)
    best = None
    for rep in range(6):
        res = bs.msm(d.ptr.value, n=n, on_device=True); t = pkg.msm_last_timing()
        if rep and (best is None or t["total_ms"] < best["total_ms"]): best = t
    want = pkg.synth_expected_msm(curve, group, 42, sc)
    ok = None
    if want is not None:
        ok = bool(np.array_equal(pkg.point_to_affine(curve, group, res), pkg.point_to_affine(curve, group, want)))
    bs.close(); d.close()
    print(json.dumps({"flow": tag, "curve": curve, "group": group, "log2_n": logn, "ok": ok, **{k: round(v, 3) for k, v in best.items()}}), flush=True)
PY
REPO=$R TAG=lists BIG=1 python /tmp/flow.py > $O/flow_lists.txt 2>&1; echo "rc=$?"
cut -c1-220 $O/flow_lists.txt
cat > /tmp/skew.py <<'PY'
import json, os, sys
sys.path.insert(0, os.environ["REPO"])
import numpy as np
from __graft_entry__ import load_package
pkg = load_package(); pkg.init(0)
tag = os.environ.get("TAG", "default")
def timed(curve, group, pts, sc, name):
    n = len(sc)
    bs = pkg.BaseSet(curve, group, pts); d = pkg.DeviceBuffer.from_numpy(sc)
    best = None
    for rep in range(5):
        res = bs.msm(d.ptr.value, n=n, on_device=True); t = pkg.msm_last_timing()
        if rep and (best is None or t["total_ms"] < best["total_ms"]): best = t
    ok = bool(np.array_equal(pkg.point_to_affine(curve, group, res), pkg.point_to_affine(curve, group, pkg.synth_expected_msm(curve, group, 42, sc))))
    bs.close(); d.close()
    print(json.dumps({"merge": tag, "case": name, "ok": ok, **{k: round(v, 3) for k, v in best.items()}}), flush=True)
for curve, group, logn in ((0, 1, 20), (0, 1, 15), (1, 1, 12), (1, 2, 15), (0, 2, 17)):
    n = 1 << logn
    pts = pkg.synth_points(curve, group, 42, n); sc = pkg.synth_scalars(curve, 43, n)
    timed(curve, group, pts, sc, f"curve {curve} G{group} 2^{logn} uniform")
    half = sc.copy(); half[::2] = pkg.api.mont_one(curve)
    timed(curve, group, pts, half, f"curve {curve} G{group} 2^{logn} half ones")
    if logn <= 17:
        same = np.tile(sc[7], (n, 1))
        timed(curve, group, pts, same, f"curve {curve} G{group} 2^{logn} all equal")
PY
REPO=$R TAG="lists" python /tmp/skew.py > $O/merge_lists.txt 2>&1; echo "rc=$?"
cut -c1-220 $O/merge_lists.txt
