#!/usr/bin/env python3
"""Window width of the small multi-scalar multiplications once the narrow reduction steps run on lane groups (msm_flow.hip.h): a
halving step costs 30 - 55 us instead of 39 - 200, so the widths chosen in round 3 (18 bits for the base fields whatever the size,
14 for Fq3) are no longer obviously right.  MNT753_MSM_PRE_C is read when the base set is created; every result is checked through
the discrete logs of the synthetic bases.   python tools/experiments/flow_window_sweep.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
pkg.init(0)


def run(curve, group, logn, c):
    os.environ.pop("MNT753_MSM_PRE_C", None)
    if c: os.environ["MNT753_MSM_PRE_C"] = str(c)
    n = 1 << logn
    pts = pkg.synth_points(curve, group, 42, n)
    sc = pkg.synth_scalars(curve, 43, n)
    bs = pkg.BaseSet(curve, group, pts)
    d = pkg.DeviceBuffer.from_numpy(sc)
    best = None
    for rep in range(5):
        res = bs.msm(d.ptr.value, n=n, on_device=True)
        t = pkg.msm_last_timing()
        if rep and (best is None or t["total_ms"] < best["total_ms"]): best = t
    ok = bool(np.array_equal(pkg.point_to_affine(curve, group, res), pkg.point_to_affine(curve, group, pkg.synth_expected_msm(curve, group, 42, sc))))
    plan = pkg.msm_last_plan()
    bs.close(); d.close()
    print(json.dumps({"curve": curve, "group": group, "log2_n": logn, "pre_c": c, "c": plan["window_bits"], "T": plan["entries_per_lane"], "pair": plan["pair_levels"],
                      "irr": plan["irr_levels"], "ok": ok, **{k: round(v, 3) for k, v in best.items()}}), flush=True)
    assert ok


for curve, group, sizes, widths in ((1, 1, (12, 13, 14, 15), (None, 10, 12, 13, 14, 15, 16)), (1, 2, (12, 13, 14, 15), (None, 10, 11, 12, 13, 15, 16)),
                                    (0, 1, (14, 15, 17), (None, 12, 14, 16)), (0, 2, (14, 15, 17), (None, 10, 12, 14, 16))):
    for logn in sizes:
        for c in widths:
            run(curve, group, logn, c)
