#!/usr/bin/env python3
"""Latency sweep of the small multi-scalar multiplications (BASELINE config 5 at 1 .. 8 GPUs: MNT6753, 2^12 .. 2^15 points per device):
window width of the table (MNT753_MSM_PRE_C, read when the base set is created), regular / irregular pairing levels
(MNT753_MSM_PAIR / MNT753_MSM_IRR) and the floor of entries per accumulate lane (MNT753_MSM_TMIN).  One JSON line per configuration;
every result is checked through the discrete logs of the synthetic bases.   python tools/experiments/small_msm_sweep.py [quick]"""
import itertools
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
pkg.init(0)


def run(curve, group, logn, env):
    for k in ("MNT753_MSM_PRE_C", "MNT753_MSM_PAIR", "MNT753_MSM_IRR", "MNT753_MSM_TMIN", "MNT753_MSM_ROUNDS"):
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    n = 1 << logn
    pts = pkg.synth_points(curve, group, 42, n)
    sc = pkg.synth_scalars(curve, 43, n)
    bs = pkg.BaseSet(curve, group, pts)
    d = pkg.DeviceBuffer.from_numpy(sc)
    best, ph = None, None
    for rep in range(4):
        res = bs.msm(d.ptr.value, n=n, on_device=True)
        t = pkg.msm_last_timing()
        if rep and (best is None or t["total_ms"] < best):
            best, ph = t["total_ms"], t
    ok = bool(np.array_equal(pkg.point_to_affine(curve, group, res), pkg.point_to_affine(curve, group, pkg.synth_expected_msm(curve, group, 42, sc))))
    plan = pkg.msm_last_plan()
    bs.close(); d.close()
    print(json.dumps({"curve": curve, "group": group, "log2_n": logn, "env": env, "ms": round(best, 3), "sort_ms": round(ph["sort_ms"], 3),
                      "accumulate_ms": round(ph["accumulate_ms"], 3), "reduce_ms": round(ph["reduce_ms"], 3), "c": plan["window_bits"],
                      "T": plan["entries_per_lane"], "pair": plan["pair_levels"], "irr": plan["irr_levels"], "ok": ok}), flush=True)
    assert ok


quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
for curve in ((1,) if quick else (1, 0)):
    sizes = (12, 13, 15) if curve == 1 else (14, 17)
    for logn in sizes:
        # G1
        for c in (None, 12, 14, 16):
            for tmin in (None, 4, 2):
                env = {}
                if c: env["MNT753_MSM_PRE_C"] = c
                if tmin: env["MNT753_MSM_TMIN"] = tmin
                run(curve, 1, logn, env)
        # G2
        for c in ((None, 12, 16) if curve == 1 else (None,)):
            for irr in (None, 1, 2, 3):
                for tmin in (None, 4):
                    env = {}
                    if c: env["MNT753_MSM_PRE_C"] = c
                    if irr: env["MNT753_MSM_IRR"] = irr
                    if tmin: env["MNT753_MSM_TMIN"] = tmin
                    run(curve, 2, logn, env)
