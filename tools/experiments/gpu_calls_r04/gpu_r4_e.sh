#!/bin/bash
# round 4, GPU call E: RCCL fold inside the boundary (tests), kernel traces of the two edge merges on a skewed vector, Fq3 pair-lanes reduce
mkdir -p gpurun_out/r4e
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4e
R=$PWD
( time python -m pytest tests/test_rccl_gpu.py tests/test_prover_gpu.py -m gpu -x -q -s ) > $O/pytest.log 2>&1
echo "pytest rc=$?"; grep -E "RCCL|exchange|passed|failed|error" $O/pytest.log | tail -8
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
REPO=$R MNT753_EDGE_TREE=0 rocprofv3 --kernel-trace --stats -d $O/kt_old -o t -- python3 /tmp/skew1.py > $O/skew_old.log 2>&1
cd $R
python3 - <<'PY'
import sqlite3, glob, os, collections
O = os.path.join(os.getcwd(), "gpurun_out", "r4e")
for tag in ("kt_tree", "kt_old"):
    for db in glob.glob(f"{O}/{tag}/**/*_results.db", recursive=True):
        con = sqlite3.connect(db); cur = con.cursor()
        t = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
        kd = [x for x in t if "kernel_dispatch" in x][0]; ks = [x for x in t if "kernel_symbol" in x][0]
        agg = collections.defaultdict(list)
        for n, dt in cur.execute(f"select s.display_name, d.end - d.start from {kd} d join {ks} s on d.kernel_id = s.id"):
            agg[n.split("(")[0].replace("void mnt753::", "")].append(dt)
        with open(f"{O}/{tag}_stats.txt", "w") as f:
            for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
                if "precompute" in n: continue
                f.write(f"{n[:60]:60s} calls {len(v):4d} avg_us {sum(v)/len(v)/1e3:9.1f} max_us {max(v)/1e3:9.1f} total_ms {sum(v)/1e6:8.2f}\n")
        con.close(); os.remove(db)
    print("==", tag); print(open(f"{O}/{tag}_stats.txt").read()[:1800])
PY
python - > $O/fq3_pair3.txt 2>&1 <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from __graft_entry__ import load_package
pkg = load_package(); pkg.init(0)
for logn in (12, 13, 14, 15):
    n = 1 << logn
    pts = pkg.synth_points(1, 2, 42, n); sc = pkg.synth_scalars(1, 43, n)
    bs = pkg.BaseSet(1, 2, pts); d = pkg.DeviceBuffer.from_numpy(sc)
    best = None
    for rep in range(5):
        res = bs.msm(d.ptr.value, n=n, on_device=True); t = pkg.msm_last_timing()
        if rep and (best is None or t["total_ms"] < best["total_ms"]): best = t
    ok = bool(np.array_equal(pkg.point_to_affine(1, 2, res), pkg.point_to_affine(1, 2, pkg.synth_expected_msm(1, 2, 42, sc))))
    bs.close(); d.close()
    print(json.dumps({"pair3": os.environ.get("MNT753_REDUCE_PAIR3", "0"), "log2_n": logn, **{k: round(v, 3) for k, v in best.items()}, "ok": ok}), flush=True)
PY
cp $O/fq3_pair3.txt $O/fq3_pair3_off.txt
MNT753_REDUCE_PAIR3=1 python - > $O/fq3_pair3_on.txt 2>&1 <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from __graft_entry__ import load_package
pkg = load_package(); pkg.init(0)
for logn in (12, 13, 14, 15):
    n = 1 << logn
    pts = pkg.synth_points(1, 2, 42, n); sc = pkg.synth_scalars(1, 43, n)
    bs = pkg.BaseSet(1, 2, pts); d = pkg.DeviceBuffer.from_numpy(sc)
    best = None
    for rep in range(5):
        res = bs.msm(d.ptr.value, n=n, on_device=True); t = pkg.msm_last_timing()
        if rep and (best is None or t["total_ms"] < best["total_ms"]): best = t
    ok = bool(np.array_equal(pkg.point_to_affine(1, 2, res), pkg.point_to_affine(1, 2, pkg.synth_expected_msm(1, 2, 42, sc))))
    bs.close(); d.close()
    print(json.dumps({"pair3": os.environ.get("MNT753_REDUCE_PAIR3", "0"), "log2_n": logn, **{k: round(v, 3) for k, v in best.items()}, "ok": ok}), flush=True)
PY
cat $O/fq3_pair3_off.txt $O/fq3_pair3_on.txt
