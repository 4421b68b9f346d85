"""Development check (GPU box): MSM time and correctness on skewed scalar distributions."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package(); pkg.init(0)
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 18
n = 1 << logn
pts = pkg.synth_points(0, 1, 42, n)
bs = pkg.BaseSet(0, 1, pts)
rnd = pkg.synth_scalars(0, 43, n)
one = pkg.api.mont_one(0)
cases = {"uniform": rnd}
eq = np.tile(rnd[0], (n, 1)); cases["all equal"] = eq
z = np.zeros_like(rnd); z[::2] = one; z[1::16] = rnd[1::16]; cases["half ones, 6% dense, rest zero"] = z
small = rnd.copy(); cases["two values"] = np.where((np.arange(n) % 2)[:, None] == 0, rnd[0], rnd[1])
for name, sc in cases.items():
    d = pkg.DeviceBuffer.from_numpy(np.ascontiguousarray(sc))
    out = bs.msm(d.ptr.value, n=n, on_device=True)
    t = time.time(); out = bs.msm(d.ptr.value, n=n, on_device=True); dt = time.time() - t
    ok = np.array_equal(pkg.point_to_affine(0, 1, out), pkg.point_to_affine(0, 1, pkg.synth_expected_msm(0, 1, 42, np.ascontiguousarray(sc))))
    print(f"2^{logn} {name:34s} {dt*1e3:8.1f} ms ok={ok} " + " ".join(f"{k}={v:.1f}" for k, v in pkg.msm_last_timing().items()), flush=True)
