#!/bin/bash
# round 4, GPU call F: the simplified binary tree level against pointer jumping on uniform and skewed vectors (timings + kernel traces)
mkdir -p gpurun_out/r4f
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4f
R=$PWD
( time python -m pytest tests/test_msm_gpu.py -m gpu -x -q ) > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -2 $O/pytest.log
cat > /tmp/skew.py <<'PY'
import json, os, sys
sys.path.insert(0, os.environ["REPO"])
import numpy as np
from __graft_entry__ import load_package
pkg = load_package(); pkg.init(0)
tag = "tree" if os.environ.get("MNT753_EDGE_TREE", "1") != "0" else "pointer-jumping"
def timed(curve, group, pts, sc, name):
    n = len(sc)
    bs = pkg.BaseSet(curve, group, pts); d = pkg.DeviceBuffer.from_numpy(sc)
    best = None
    for rep in range(5):
        res = bs.msm(d.ptr.value, n=n, on_device=True); t = pkg.msm_last_timing()
        if rep and (best is None or t["total_ms"] < best["total_ms"]): best = t
    bs.close(); d.close()
    print(json.dumps({"merge": tag, "case": name, **{k: round(v, 3) for k, v in best.items()}}), flush=True)
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
REPO=$R python /tmp/skew.py > $O/merge_tree.txt 2>&1; echo "tree rc=$?"
REPO=$R MNT753_EDGE_TREE=0 python /tmp/skew.py > $O/merge_old.txt 2>&1; echo "old rc=$?"
paste -d'\n' $O/merge_tree.txt $O/merge_old.txt | cut -c1-200
cat > /tmp/skew1.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["REPO"])
import numpy as np
from __graft_entry__ import load_package
pkg = load_package(); pkg.init(0)
n = 1 << 15
pts = pkg.synth_points(0, 1, 42, n); sc = pkg.synth_scalars(0, 43, n)
sc[::2] = pkg.api.mont_one(0)
bs = pkg.BaseSet(0, 1, pts); d = pkg.DeviceBuffer.from_numpy(sc)
for rep in range(4):
    bs.msm(d.ptr.value, n=n, on_device=True)
print(pkg.msm_last_timing())
PY
cd /tmp
REPO=$R rocprofv3 --kernel-trace --stats -d $O/kt_tree -o t -- python3 /tmp/skew1.py > $O/skew_tree.log 2>&1
cd $R
python3 - <<'PY'
import sqlite3, glob, os, collections
O = os.path.join(os.getcwd(), "gpurun_out", "r4f")
for tag in ("kt_tree",):
    for db in glob.glob(f"{O}/{tag}/**/*_results.db", recursive=True):
        con = sqlite3.connect(db); cur = con.cursor()
        t = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
        kd = [x for x in t if "kernel_dispatch" in x][0]; ks = [x for x in t if "kernel_symbol" in x][0]
        rows = list(cur.execute(f"select s.display_name, d.end - d.start, d.start from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
        lv = [round(dt / 1e3, 1) for n, dt, _ in rows if "k_edge_tree_level" in n]
        print("k_edge_tree_level durations (us), in launch order, last MSM:", lv[-16:])
        con.close(); os.remove(db)
PY
