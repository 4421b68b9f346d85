#!/usr/bin/env python3
"""SURVEY.md section 8e, plan B, sized on ONE GPU: partition an MSM over N devices by WINDOWS instead of by points.

Plan A (built): device g runs the whole Pippenger on the slice [g n / N, (g + 1) n / N) of the points: W n / N list entries, the same
2^(c-1) buckets as the full MSM (the fixed phases -- sort, edge merge, bucket reduction -- do not shrink).
Plan B: every device keeps the window table of ALL n points (9.4 GB per 2^20 G1 points; it fits) and takes W / N of the W windows: the same
W n / N entries, the same bucket set, gathers spread over the whole table instead of 1 / N of it.

With the window table every window indexes one bucket set, so "device g's windows" is simply: only the digits of windows
[g W / N, (g + 1) W / N) are non-zero.  That needs no kernel change to measure: a scalar below 2^(c W / N - 1) has exactly those digits
for g = 0, so plan B's device 0 is the MSM of all n points with short scalars.  (The sort stage still scans all W windows of every
scalar here; a built plan B would scan W / N of them, so its sort time is an upper bound.)   One JSON line per size."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
pkg.init(0)
R_MOD = {0: 0x0001c4c62d92c41110229022eee2cdadb7f997505b8fafed5eb7e8f96c97d87307fdb925e8a0ed8d99d124d9a15af79db26c5c28c859a99b3eebca9429212636b9dff97634993aa4d6c381bc3f0057974ea099170fa13a4fd90776e240000001,
         1: 0x0001c4c62d92c41110229022eee2cdadb7f997505b8fafed5eb7e8f96c97d87307fdb925e8a0ed8d99d124d9a15af79db117e776f218059db80f0da5cb537e38685acce9767254a4638810719ac425f0e39d54522cdd119f5e9063de245e8001}


def short_scalars(curve, n, bits, seed):
    """n uniform integers below 2^bits, in the wire (Montgomery, R = 2^768) form"""
    rng = np.random.default_rng(seed)
    r = R_MOD[curve]
    out = np.zeros((n, 12), dtype=np.uint64)
    raw = rng.integers(0, 1 << 62, size=(n, (bits + 61) // 62), dtype=np.uint64)
    for i in range(n):
        v = 0
        for k, x in enumerate(raw[i]):
            v |= int(x) << (62 * k)
        v &= (1 << bits) - 1
        m = (v << 768) % r
        for k in range(12):
            out[i, k] = (m >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return out


def time_msm(bs, sc, n):
    d = pkg.DeviceBuffer.from_numpy(sc)
    best = None
    for rep in range(4):
        res = bs.msm(d.ptr.value, n=n, on_device=True)
        t = pkg.msm_last_timing()
        if rep and (best is None or t["total_ms"] < best["total_ms"]):
            best = t
    d.close()
    return res, best


for curve, group, logn in ((0, 1, 20), (1, 1, 15)):
    n = 1 << logn
    pts = pkg.synth_points(curve, group, 42, n)
    bs = pkg.BaseSet(curve, group, pts)
    full = pkg.synth_scalars(curve, 43, n)
    _, t_full = time_msm(bs, full, n)
    plan = pkg.msm_last_plan()
    c, W = plan["window_bits"], plan["windows"]
    for N in (2, 4, 8):
        wn = (W + N - 1) // N
        bits = c * wn - 1
        sc = short_scalars(curve, n, bits, 7 + N)
        res, t_b = time_msm(bs, sc, n)
        ok = bool(np.array_equal(pkg.point_to_affine(curve, group, res), pkg.point_to_affine(curve, group, pkg.synth_expected_msm(curve, group, 42, sc))))
        # plan A on the same GPU: the first n / N points with full scalars (a base set of its own: the slice's table is 1 / N of the memory)
        k = n // N
        bs_a = pkg.BaseSet(curve, group, pts[:k])
        _, t_a = time_msm(bs_a, full[:k], k)
        plan_a = pkg.msm_last_plan()
        bs_a.close()
        print(json.dumps({"curve": curve, "log2_n": logn, "N": N, "c": c, "W": W, "windows_per_device": wn,
                          "plan_B_ms": round(t_b["total_ms"], 3), "plan_B_phases": {k2: round(v, 3) for k2, v in t_b.items()},
                          "plan_A_ms": round(t_a["total_ms"], 3), "plan_A_phases": {k2: round(v, 3) for k2, v in t_a.items()}, "plan_A_c": plan_a["window_bits"],
                          "full_ms": round(t_full["total_ms"], 3), "parity_ok": ok}), flush=True)
        assert ok
    bs.close()
