#!/usr/bin/env python3
"""Pairing levels for the base-field sets between 2^15 and 2^17 points (and 3 x 2^15: the MNT6753 prover's H | L | B1), where the
accumulate walk is 21 .. 84 mixed additions deep per lane: do one or two batched-affine levels pay now?  MNT753_MSM_PAIR / _IRR are
read when the base set is created and per call.   python tools/experiments/pair_levels_small_sweep.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
pkg.init(0)

for curve, n in ((1, 1 << 15), (1, 3 << 15), (0, 1 << 15), (0, 1 << 16), (0, 3 << 15), (0, 1 << 17)):
    pts = pkg.synth_points(curve, 1, 42, n)
    sc = pkg.synth_scalars(curve, 43, n)
    want = pkg.point_to_affine(curve, 1, pkg.synth_expected_msm(curve, 1, 42, sc))
    for pair, irr in ((None, None), (1, 0), (1, 1), (2, 0), (2, 1), (3, 0)):
        for k in ("MNT753_MSM_PAIR", "MNT753_MSM_IRR"): os.environ.pop(k, None)
        if pair is not None: os.environ["MNT753_MSM_PAIR"] = str(pair); os.environ["MNT753_MSM_IRR"] = str(irr)
        bs = pkg.BaseSet(curve, 1, pts)
        d = pkg.DeviceBuffer.from_numpy(sc)
        best = None
        for rep in range(5):
            res = bs.msm(d.ptr.value, n=n, on_device=True)
            t = pkg.msm_last_timing()
            if rep and (best is None or t["total_ms"] < best["total_ms"]): best = t
        ok = bool(np.array_equal(pkg.point_to_affine(curve, 1, res), want))
        plan = pkg.msm_last_plan()
        bs.close(); d.close()
        print(json.dumps({"curve": curve, "n": n, "pair_env": pair, "irr_env": irr, "pair": plan["pair_levels"], "irr": plan["irr_levels"], "T": plan["entries_per_lane"],
                          "c": plan["window_bits"], "ok": ok, **{k: round(v, 3) for k, v in best.items()}}), flush=True)
        assert ok
