import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from __graft_entry__ import load_package
import oracle_lib as O
pkg = load_package(); pkg.init(0)
def run(curve, group, n, c):
    pts = pkg.synth_points(curve, group, 7, n); sc = pkg.synth_scalars(curve, 8, n)
    pkg.lib().mnt753_msm_set_window_bits(c)
    bs = pkg.BaseSet(curve, group, pts)
    got = pkg.point_to_affine(curve, group, bs.msm(sc))
    ok = np.array_equal(got, O.msm(curve, group, pts, sc))
    return "ok" if ok else "BAD"
for curve, group in ((0, 1), (0, 2), (1, 2)):
    for n in (4, 5, 6, 8, 16, 40):
        print(curve, group, n, " ".join(f"c{c}:{run(curve, group, n, c)}" for c in (2, 3, 5, 8, 12)), flush=True)
