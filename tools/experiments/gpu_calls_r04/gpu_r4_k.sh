#!/bin/bash
# round 4, GPU call K: list-driven edge levels on lane groups: parity, skewed vectors, trace
mkdir -p gpurun_out/r4k
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4k
R=$PWD
( time python -m pytest tests/test_device_kat_gpu.py tests/test_msm_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^FAILED|passed|failed" $O/pytest.log | cut -c1-150 | head -20
for env in "MNT753_MSM_TMIN=1" "MNT753_MSM_TMIN=2 MNT753_EDGE_FLOW_NODES=100000000" "MNT753_EDGE_FLOW_NODES=100000000" "MNT753_EDGE_FLOW_NODES=0"; do
  echo "== $env"
  env $env python -m pytest tests/test_msm_gpu.py -m gpu -q -k "test_golden or skewed or edge_cases or cancellations or randomized" 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-120
done > $O/golden_matrix.log 2>&1
cat $O/golden_matrix.log
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
REPO=$R TAG="tree+groups(list)" python /tmp/skew.py > $O/merge_tree_flow.txt 2>&1; echo "rc=$?"
cut -c1-220 $O/merge_tree_flow.txt
cat > /tmp/one.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["REPO"])
from __graft_entry__ import load_package
pkg = load_package(); pkg.init(0)
n = 1 << 15
pts = pkg.synth_points(1, 2, 42, n); sc = pkg.synth_scalars(1, 43, n)
bs = pkg.BaseSet(1, 2, pts); d = pkg.DeviceBuffer.from_numpy(sc)
for rep in range(4): bs.msm(d.ptr.value, n=n, on_device=True)
print(pkg.msm_last_timing(), pkg.msm_last_plan())
PY
cd /tmp
REPO=$R rocprofv3 --kernel-trace --stats -d $O/kt -o t -- python3 /tmp/one.py > $O/one.log 2>&1
cd $R
python3 - <<'PY'
import sqlite3, glob, os
O = os.path.join(os.getcwd(), "gpurun_out", "r4k")
for db in glob.glob(f"{O}/kt/**/*_results.db", recursive=True):
    con = sqlite3.connect(db); cur = con.cursor()
    t = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in t if "kernel_dispatch" in x][0]; ks = [x for x in t if "kernel_symbol" in x][0]
    rows = list(cur.execute(f"select s.display_name, d.end - d.start, d.grid_size_x, d.start from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    last = max(i for i, r in enumerate(rows) if "k_scalar_digits" in r[0])
    with open(f"{O}/mnt6_g2_2p15_last_msm_kernels.txt", "w") as f:
        for n, dt, g, st in rows[last:]:
            f.write(f"{dt / 1e3:9.1f} us  grid {g:8d}  {n.split('(')[0][:100]}\n")
    con.close(); os.remove(db)
print(open(f"{O}/mnt6_g2_2p15_last_msm_kernels.txt").read())
PY
