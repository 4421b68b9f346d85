"""Development check (GPU box): do MSMs started on different base sets overlap?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package(); pkg.init(0)
curve = int(os.environ.get("CURVE", 1)); logn = int(sys.argv[1]) if len(sys.argv) > 1 else 15
n = 1 << logn
sc = pkg.synth_scalars(curve, 5, n); d = pkg.DeviceBuffer.from_numpy(sc)
sets = [pkg.BaseSet(curve, 1, pkg.synth_points(curve, 1, 10 + k, n)) for k in range(4)] + [pkg.BaseSet(curve, 2, pkg.synth_points(curve, 2, 20, n))]
for b in sets: b.msm(d.ptr.value, n=n, on_device=True)          # warm (workspaces)
t = time.time()
for b in sets: b.msm(d.ptr.value, n=n, on_device=True)
seq = time.time() - t
per = []
for b in sets:
    t1 = time.time(); b.msm(d.ptr.value, n=n, on_device=True); per.append((time.time() - t1) * 1e3)
t = time.time()
for b in sets: b.msm_start(d.ptr.value, n)
t_launch = time.time() - t
for b in sets: b.msm_finish()
con = time.time() - t
print(f"curve={curve} n=2^{logn}: sequential {seq*1e3:.1f} ms (each: {' '.join('%.1f' % x for x in per)}), concurrent {con*1e3:.1f} ms (launch calls returned after {t_launch*1e3:.1f} ms)")
