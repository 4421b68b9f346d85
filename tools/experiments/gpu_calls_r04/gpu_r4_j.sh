#!/bin/bash
# round 4, GPU call J: kernel trace of the MNT6753 G2 2^15 MSM with the edge levels on lane groups
mkdir -p gpurun_out/r4j
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4j
R=$PWD
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
tail -1 $O/one.log
python3 - <<'PY'
import sqlite3, glob, os
O = os.path.join(os.getcwd(), "gpurun_out", "r4j")
for db in glob.glob(f"{O}/kt/**/*_results.db", recursive=True):
    con = sqlite3.connect(db); cur = con.cursor()
    t = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in t if "kernel_dispatch" in x][0]; ks = [x for x in t if "kernel_symbol" in x][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    rows = list(cur.execute(f"select s.display_name, d.end - d.start, d.grid_size_x, d.start from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    last = max(i for i, r in enumerate(rows) if "k_scalar_digits" in r[0])
    prev = None
    for n, dt, g, st in rows[last:]:
        gap = (st - prev) / 1e3 if prev else 0.0
        prev = st + dt
        print(f"{dt / 1e3:9.1f} us  gap {gap:6.1f}  grid {g:8d}  {n.split('(')[0][:100]}")
    con.close(); os.remove(db)
PY
