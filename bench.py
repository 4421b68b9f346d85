#!/usr/bin/env python3
"""bench.py -- headline benchmark: MNT4753 G1 Pippenger MSM, 2^20 bases per GPU (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One process per GPU.  A step is ONE multi-scalar multiplication over the rank's 2^20 (base, scalar) pairs with the
bases and scalars already resident in HBM, followed -- when N > 1 -- by the only exchange the path has: an
all_gather (RCCL) of one 288-byte projective point per rank and the serial fold of the N partial sums.  Per-GPU work
is fixed (weak scaling): N ranks compute an N * 2^20-point MSM.  Rank 0 prints ONE JSON line.

PyTorch is plumbing here (device memory for the scalars, the stream handle, torch.distributed); every kernel is in
libmnt753_hip.so, reached through the C ABI.  The oracle is used only for the cpu_baseline leg.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_N = 20
ALGO_BYTES_PER_PAIR = 288          # 192 B affine G1 base + 96 B scalar, read once
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: 8 TB/s HBM3E
MODMUL_PEAK_PER_S = 22.0e9         # measured chip peak of the 753-bit Montgomery multiplier (profiles/r01/mulbench_mi355x.txt)


def cpu_baseline(pkg, pts, sc):
    """The oracle's chunked BDLO12 Pippenger (restatement of libff multi_exp, all host threads) on a bounded
    sample of the same workload: the first 2^15 pairs."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    threads = O.lib().oracle_max_threads()
    n = 1 << 12
    t0 = time.time(); O.msm(0, 1, pts[:n], sc[:n], chunks=threads); t_small = time.time() - t0
    # scale the sample towards ~12 s of CPU work, capped at 2^19 pairs
    n2 = n
    while n2 < (1 << 19) and t_small * (n2 * 2 / n) < 15.0:
        n2 *= 2
    t0 = time.time(); got = O.msm(0, 1, pts[:n2], sc[:n2], chunks=threads); dt = time.time() - t0
    return dict(value=n2 / dt, unit="points/s", cores=threads, kind="port",
                sample=f"first 2^{n2.bit_length() - 1} (base, scalar) pairs of the benchmark input, oracle chunked BDLO12, {dt:.1f} s"), got, n2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=LOG_N, help="log2 of the bases per GPU (benchmark config: 20)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    pkg = load_package()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # BENCH_SHARE_GPU=1 (development only): all ranks share GPU 0 and exchange over gloo, to exercise the N > 1 flow
    # on a single-GPU box; the real multi-GPU run is one rank per GPU over RCCL.
    share = os.environ.get("BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    pkg.init(dev_index)

    n = 1 << args.log_n
    # rank g owns slice g of an (world * n)-point MSM: distinct seeds per rank
    pts = pkg.synth_points(0, 1, 42 + 1000 * rank, n)
    sc = pkg.synth_scalars(0, 43 + 1000 * rank, n)
    bases = pkg.BaseSet(0, 1, pts)                                  # parameters: resident before timing (main.cpp:201-203)
    d_sc = torch.from_numpy(sc.view(np.int64)).to(device)          # scalars resident in HBM
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        local = bases.msm(d_sc.data_ptr(), n=n, on_device=True, stream=stream)
        return pkg.parallel.msm_sharded(pkg.api, 0, 1, local, None if share else device)

    for _ in range(args.warmup):
        out = step()
    acc_ms, tot_ms = [], []
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
        tm = pkg.msm_last_timing()
        acc_ms.append(tm["accumulate_ms"]); tot_ms.append(tm)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # parity of what was just timed: every rank's slice through its discrete logs, folded like the timed path
    exp_local = pkg.synth_expected_msm(0, 1, 42 + 1000 * rank, sc)
    exp = pkg.parallel.msm_sharded(pkg.api, 0, 1, exp_local, None if share else device)
    ok = bool(np.array_equal(pkg.point_to_affine(0, 1, out), pkg.point_to_affine(0, 1, exp)))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * n * args.steps / elapsed
        acc = float(np.mean(acc_ms))
        achieved = ALGO_BYTES_PER_PAIR * n / (acc * 1e-3) / 1e9
        plan = pkg.msm_last_plan()
        plan_c, windows = plan["window_bits"], plan["windows"]
        levels = plan.get("pair_levels", 0)
        # Montgomery products per sorted entry: 6 per affine pair addition (3 + 3 for the simultaneous inversion) on 1/2, 1/4, ...
        # of the entries, ~94 / batch for the divstep inversion, 11 per mixed addition on what is left
        prod_per_entry = sum((6.0 + 94.0 / (155.0 if l == 1 else 96.0)) / 2 ** l for l in range(1, levels + 1)) + 11.0 / 2 ** levels
        kernel_name = "k_bucket_accumulate<Mnt4G1>" if levels == 0 else f"k_pair_add<Mnt4G1> x{levels} + k_bucket_accumulate<Mnt4G1>"
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01", "accumulate_traffic.json")
        if os.path.exists(tpath) and args.log_n == LOG_N:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        line = {
            "metric": "G1 MSM points/sec at 2^20 (MNT4753)",
            "value": value,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "parity_ok": ok,
            "config": {"workload": f"MNT4753 G1 Pippenger MSM, 2^{args.log_n} bases per GPU, bit-exact vs libff::multi_exp",
                       "curve": "MNT4753", "group": "G1", "points_per_gpu": n, "window_bits": plan_c, "windows": windows,
                       "window_table": plan["window_table"],
                       "parallelism": f"slice-per-gpu x{world}, all_gather of one projective point per rank"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "kernel": kernel_name, "kernel_ms": acc,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_PAIR * n,
                         "note": "bucket accumulation phase of one MSM (HIP events on the launch stream): integer-ALU bound in the "
                                 "projective accumulate, scattered-HBM-sector bound in the first pairing level; far from the streaming HBM roof either way",
                         "pair_levels": levels, "products_per_entry": prod_per_entry,
                         "modmul_per_s": prod_per_entry * windows * n / (acc * 1e-3), "modmul_peak_per_s": MODMUL_PEAK_PER_S,
                         "modmul_frac": prod_per_entry * windows * n / (acc * 1e-3) / MODMUL_PEAK_PER_S,
                         "mixed_addition_equivalents_per_s": windows * n / (acc * 1e-3)},
            "phases_ms": {k: float(np.mean([t[k] for t in tot_ms])) for k in tot_ms[0]},
        }
        if world == 1 and not args.no_cpu_baseline:
            base, got, n2 = cpu_baseline(pkg, pts, sc)
            chk = pkg.BaseSet(0, 1, pts[:n2])
            same = bool(np.array_equal(pkg.point_to_affine(0, 1, chk.msm(sc[:n2])), got))
            base["matches_gpu_on_sample"] = same
            line["cpu_baseline"] = base
            line["parity_ok"] = ok and same
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("bench.py: PARITY FAILURE")


if __name__ == "__main__":
    main()
