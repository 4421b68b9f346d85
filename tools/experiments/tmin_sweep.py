#!/usr/bin/env python3
"""Floor of entries per accumulate lane once the levels of the edge merge run on lane groups (a level costs 30 - 60 us instead of 75 - 310):
shorter lanes mean a shorter walk and a deeper merge.  MNT753_MSM_TMIN is read per call.   python tools/experiments/tmin_sweep.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
pkg.init(0)

for curve, group, sizes in ((1, 1, (12, 13, 14)), (0, 1, (12, 13, 14)), (1, 2, (12, 13)), (0, 2, (12, 13, 14))):
    for logn in sizes:
        n = 1 << logn
        pts = pkg.synth_points(curve, group, 42, n)
        sc = pkg.synth_scalars(curve, 43, n)
        bs = pkg.BaseSet(curve, group, pts)
        d = pkg.DeviceBuffer.from_numpy(sc)
        want = pkg.point_to_affine(curve, group, pkg.synth_expected_msm(curve, group, 42, sc))
        for tmin in (None, 16, 6, 4, 3, 2):
            os.environ.pop("MNT753_MSM_TMIN", None)
            if tmin: os.environ["MNT753_MSM_TMIN"] = str(tmin)
            best = None
            for rep in range(5):
                res = bs.msm(d.ptr.value, n=n, on_device=True)
                t = pkg.msm_last_timing()
                if rep and (best is None or t["total_ms"] < best["total_ms"]): best = t
            ok = bool(np.array_equal(pkg.point_to_affine(curve, group, res), want))
            plan = pkg.msm_last_plan()
            print(json.dumps({"curve": curve, "group": group, "log2_n": logn, "tmin": tmin, "T": plan["entries_per_lane"], "c": plan["window_bits"], "pair": plan["pair_levels"],
                              "ok": ok, **{k: round(v, 3) for k, v in best.items()}}), flush=True)
            assert ok
        bs.close(); d.close()
