#!/usr/bin/env python3
"""GPU box: MSM time at the per-device sizes an N-way split of the benchmark configurations produces, by pairing-level count
-- the data behind pair_levels() in csrc/msm_host.hpp (the batch-length floor PAIR_MIN_B = 8 was settled by this sweep in round 3) and behind the predicted 1/2/4/8-GPU
curve in DESIGN.md section 5 (multiexp.tcc:417-440 splits ONE array into contiguous slices; slice g runs the whole Pippenger).

    python tools/slice_sweep.py [--quick] [--out gpurun_out/slice_sweep.json]

For every (curve, group, log2 n): the default plan, then MNT753_MSM_PAIR in {0, 1, 2, 3, 4}.
Every result is checked through the discrete logs of the synthetic bases (synth_expected_msm)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="default plan only (what bench.py's extras run)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "slice_sweep.json"))
    ap.add_argument("--configs", default="0:1:20,0:1:19,0:1:18,0:1:17,0:2:20,0:2:19,0:2:18,0:2:17,1:1:15,1:1:14,1:1:13,1:1:12,1:2:15,1:2:14,1:2:13,1:2:12")
    args = ap.parse_args()
    pkg = load_package()
    pkg.init(0)
    rows = []
    for cfg in args.configs.split(","):
        curve, group, size = cfg.split(":")
        curve, group = int(curve), int(group)
        # "20": 2^20 points; "n3145727": that many (the concatenated set H | L | B1 of B::groth16_C at d = 2^20 - 1)
        n = int(size[1:]) if size.startswith("n") else 1 << int(size)
        logn = int(size) if not size.startswith("n") else float(np.log2(n))
        pts = pkg.synth_points(curve, group, 42, n)
        sc = pkg.synth_scalars(curve, 43, n)
        exp = pkg.point_to_affine(curve, group, pkg.synth_expected_msm(curve, group, 42, sc))
        bs = pkg.BaseSet(curve, group, pts)
        dsc = pkg.DeviceBuffer.from_numpy(sc)
        variants = [(None, None)]
        if not args.quick:
            variants += [(lv, None) for lv in (0, 1, 2, 3, 4)]
        for lv, mb in variants:
            for k, v in (("MNT753_MSM_PAIR", lv),):
                if args.quick:
                    break            # --quick: whatever the environment says (tools/experiments/*.sh set the knobs themselves)
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = str(v)
            best, ok = None, True
            for rep in range(4):
                t0 = time.perf_counter()
                out = bs.msm(dsc.ptr.value, n=n, on_device=True)
                wall = (time.perf_counter() - t0) * 1e3
                tm = pkg.msm_last_timing()
                if rep and (best is None or tm["total_ms"] < best["total_ms"]):
                    best = dict(tm, wall_ms=wall)
                ok = ok and bool(np.array_equal(pkg.point_to_affine(curve, group, out), exp))
            plan = pkg.msm_last_plan()
            row = dict(curve=curve, group=group, log2_n=logn, forced_levels=lv, min_b=mb, levels=plan["pair_levels"], irr_levels=plan.get("irr_levels", 0), window_bits=plan["window_bits"],
                       table=plan["window_table"], ok=ok, **{k: round(v, 3) for k, v in best.items()})
            rows.append(row)
            print(json.dumps(row), flush=True)
        os.environ.pop("MNT753_MSM_PAIR", None)
        bs.close(); dsc.close()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(rows, f, indent=1)
    if not all(r["ok"] for r in rows):
        raise SystemExit("slice_sweep: PARITY FAILURE")


if __name__ == "__main__":
    main()
