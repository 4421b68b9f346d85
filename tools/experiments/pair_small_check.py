import os, sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
os.environ["MNT753_MSM_PAIR"] = sys.argv[1] if len(sys.argv) > 1 else "1"
os.environ["MNT753_MSM_PRECOMP"] = "1"
from __graft_entry__ import load_package
import oracle_lib as O
pkg = load_package(); pkg.init(0)
n = 600
pts = pkg.synth_points(0, 1, 5, n); sc = pkg.synth_scalars(0, 6, n)
bs = pkg.BaseSet(0, 1, pts)
got = pkg.point_to_affine(0, 1, bs.msm(sc))
print("levels", os.environ["MNT753_MSM_PAIR"], "ok", np.array_equal(got, O.msm(0, 1, pts, sc)), pkg.msm_last_plan())
