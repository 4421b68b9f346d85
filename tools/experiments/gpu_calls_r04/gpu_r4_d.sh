#!/bin/bash
# round 4, GPU call D: the whole -m gpu suite with the binary tree merge, the merge on skewed scalar vectors (old vs tree), small sizes
mkdir -p gpurun_out/r4d
export TMPDIR=/tmp
O=gpurun_out/r4d
( time python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
cat > /tmp/skew.py <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from __graft_entry__ import load_package
pkg = load_package(); pkg.init(0)
tag = "tree" if os.environ.get("MNT753_EDGE_TREE", "1") != "0" else "pointer-jumping"
def timed(curve, group, pts, sc, name):
    n = len(sc)
    bs = pkg.BaseSet(curve, group, pts); d = pkg.DeviceBuffer.from_numpy(sc)
    best = None
    for rep in range(4):
        res = bs.msm(d.ptr.value, n=n, on_device=True); t = pkg.msm_last_timing()
        if rep and (best is None or t["total_ms"] < best["total_ms"]): best = t
    bs.close(); d.close()
    print(json.dumps({"merge": tag, "case": name, **{k: round(v, 3) for k, v in best.items()}}), flush=True)
    return pkg.point_to_affine(curve, group, res)
one = pkg.api.mont_one(0)
for logn in (20, 15):
    n = 1 << logn
    pts = pkg.synth_points(0, 1, 42, n); sc = pkg.synth_scalars(0, 43, n)
    r0 = timed(0, 1, pts, sc, f"MNT4753 G1 2^{logn} uniform")
    half = sc.copy(); half[::2] = one                      # every other scalar is one: half the list lands in one bucket
    r1 = timed(0, 1, pts, half, f"MNT4753 G1 2^{logn} half ones")
    same = np.tile(sc[7], (n, 1))                           # all scalars equal: 40 buckets hold everything
    r2 = timed(0, 1, pts, same, f"MNT4753 G1 2^{logn} all equal")
    np.save(f"/tmp/skew_{tag}_{logn}.npy", np.stack([r0, r1, r2]))
n = 1 << 15
for group in (1, 2):
    pts = pkg.synth_points(1, group, 42, n); sc = pkg.synth_scalars(1, 43, n)
    timed(1, group, pts, sc, f"MNT6753 G{group} 2^15 uniform")
    half = sc.copy(); half[::2] = pkg.api.mont_one(1)
    timed(1, group, pts, half, f"MNT6753 G{group} 2^15 half ones")
for logn in (12, 13):
    n = 1 << logn
    pts = pkg.synth_points(1, 1, 42, n); sc = pkg.synth_scalars(1, 43, n)
    timed(1, 1, pts, sc, f"MNT6753 G1 2^{logn} uniform")
PY
python /tmp/skew.py > $O/merge_tree.txt 2>&1; echo "tree rc=$?"
MNT753_EDGE_TREE=0 python /tmp/skew.py > $O/merge_old.txt 2>&1; echo "old rc=$?"
python - <<'PY'
import numpy as np
for logn in (20, 15):
    a = np.load(f"/tmp/skew_tree_{logn}.npy"); b = np.load(f"/tmp/skew_pointer-jumping_{logn}.npy")
    print("same results 2^%d:" % logn, bool(np.array_equal(a, b)))
PY
paste -d'\n' $O/merge_tree.txt $O/merge_old.txt | cut -c1-200
