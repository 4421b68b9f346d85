"""Development timing (GPU box): large synthetic MSM, checked through the known discrete logs."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package(); pkg.init(0)
curve = int(os.environ.get("CURVE", 0)); group = int(os.environ.get("GROUP", 1))
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 1 << logn
t0 = time.time(); pts = pkg.synth_points(curve, group, 42, n); t1 = time.time()
sc = pkg.synth_scalars(curve, 43, n); t2 = time.time()
exp = pkg.synth_expected_msm(curve, group, 42, sc); t3 = time.time()
print(f"synth points {t1-t0:.2f}s scalars {t2-t1:.2f}s expected {t3-t2:.2f}s", flush=True)
bs = pkg.BaseSet(curve, group, pts); t4 = time.time()
print(f"bases upload+convert {t4-t3:.2f}s", flush=True)
dsc = pkg.DeviceBuffer.from_numpy(sc)
for r in range(reps):
    t = time.time()
    out = bs.msm(dsc.ptr.value, n=n, on_device=True)
    dt = time.time() - t
    tm = pkg.msm_last_timing()
    ok = np.array_equal(pkg.point_to_affine(curve, group, out), pkg.point_to_affine(curve, group, exp))
    print(f"msm 2^{logn} curve={curve} group={group}: wall {dt*1e3:.1f} ms  {n/dt/1e6:.2f} Mpts/s  ok={ok}  " +
          " ".join(f"{k}={v:.2f}" for k, v in tm.items()), flush=True)
