#!/usr/bin/env python3
"""Verdict item 5 (round 5), measured before anything is built: what would ONE multi-scalar multiplication for A AND C be worth?

Today a prove runs A's MSM over m + 1 points and C's over H | L | B1 (d + 2 m points) one after the other on the same device
(cuda_prover_piecewise.cu:70-90 runs five).  Fused, the two would be one pass over A | H | L | B1 with TWO bucket sets (the entries of
A's points go to set 0, the others to set 1): sort, pairing levels, accumulate, edge merge once; the bucket reduction twice.
Upper bound of the gain without writing it: the MSM over (1 + 3) k points as ONE set (what the fused pass costs minus the second
reduction) against the two separate MSMs.  Prints one JSON line per size; reduce_ms is what a second bucket set would add."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
pkg.init(0)


def time_msm(curve, pts, sc, seed):
    n = len(pts)
    bs = pkg.BaseSet(curve, 1, pts)
    d = pkg.DeviceBuffer.from_numpy(sc)
    best = None
    for rep in range(5):
        res = bs.msm(d.ptr.value, n=n, on_device=True)
        t = pkg.msm_last_timing()
        if rep and (best is None or t["total_ms"] < best["total_ms"]):
            best = dict(t, **pkg.msm_last_plan())
    ok = bool(np.array_equal(pkg.point_to_affine(curve, 1, res), pkg.point_to_affine(curve, 1, pkg.synth_expected_msm(curve, 1, seed, sc))))
    bs.close(); d.close()
    return best, ok


for curve, logn in ((1, 15), (0, 20)):
    k = 1 << logn
    pts = pkg.synth_points(curve, 1, 42, 4 * k)
    sc = pkg.synth_scalars(curve, 43, 4 * k)
    a, ok_a = time_msm(curve, pts[:k], sc[:k], 42)
    c, ok_c = time_msm(curve, pts[:3 * k], sc[:3 * k], 42)
    f, ok_f = time_msm(curve, pts, sc, 42)
    line = {"curve": "MNT4753" if curve == 0 else "MNT6753", "log2_k": logn, "parity_ok": ok_a and ok_c and ok_f,
            "A_k_points_ms": round(a["total_ms"], 3), "C_3k_points_ms": round(c["total_ms"], 3), "one_set_4k_points_ms": round(f["total_ms"], 3),
            "reduce_ms_of_the_4k_set": round(f["reduce_ms"], 3), "window_bits": {"A": a["window_bits"], "C": c["window_bits"], "4k": f["window_bits"]},
            "separate_ms": round(a["total_ms"] + c["total_ms"], 3),
            "fused_estimate_ms": round(f["total_ms"] + f["reduce_ms"], 3)}
    line["estimated_gain_ms"] = round(line["separate_ms"] - line["fused_estimate_ms"], 3)
    print(json.dumps(line), flush=True)
