#!/usr/bin/env python3
"""prove_mgpu.py -- the whole Groth16 prove sharded over the GPUs of one node.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        prove_mgpu.py MNT4753|MNT6753 compute <params> <input> <output> [--unfused-c]
    python prove_mgpu.py ...                      (N = 1)

Same files and same proof bytes as `./main <curve> compute ...` of the reference and as the single-GPU `main_hip`.
Decomposition = the reference's own OpenMP chunking of an MSM (multiexp.tcc:402-441) lifted to GPUs:

  * rank g keeps the slice [lo_g, hi_g) of A and of B2 resident in its HBM and, as ONE base set, its slices of H, L and B1
    concatenated (C = Ht + Lt + r Bt1 is one group element: one MSM over H_g | L_g | B1_g with the scalars h | w_L | r w instead of
    three MSMs, a scalar multiplication and two additions -- B::groth16_C of the single-process wrapper), each with its window
    table; parameter loading is outside the timed window (libsnark/main.cpp:201-203).  --unfused-c keeps the five base sets and
    the reference's five multiexps;
  * nothing of the input is read twice and nothing is funnelled through rank 0 (round 4): rank g streams from the input file the range
    of w that its slices multiply, and ranks 0 / 1 / 2 one vector of compute_H each -- ca / cb / cc (rank 0 takes cc too when there
    are only two ranks).  Each of them runs x <- cosetFFT(iFFT(x)) on its own vector (mnt753_compute_h_chain); the transformed cb and
    cc travel to rank 0 (RCCL send / recv over xGMI, 96 (d + 1) bytes each), which runs the pointwise step and the last transform
    (mnt753_compute_h_finish) and scatters slice g of coefficients_for_H to rank g (send / recv, 96 d / N bytes each).  The FFT itself
    is not sharded: 100 MB fit one GPU and a distributed NTT would move the whole vector over xGMI for ~3 % of the work;
  * the local MSMs run concurrently on their base sets' streams (mnt753_msm_start / _finish);
  * ONE all_gather per proof carries the three (five) partial points of every rank (36..108 u64 each -- latency bound);
    every rank folds them in rank order, rank 0 writes the proof.

PyTorch is plumbing: torch.distributed (backend nccl = RCCL) and the device tensors its send / recv move.  PROVE_SHARE_GPU=1
(development) lets all ranks share GPU 0 over gloo (the vectors then hop through host memory) so the flow can be exercised on a
single-GPU box.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def read_slice(path, offset_bytes, n_rows, row_words):
    return np.fromfile(path, dtype=np.uint64, count=n_rows * row_words, offset=offset_bytes).reshape(n_rows, row_words)


def main():
    if len(sys.argv) < 6 or sys.argv[2] != "compute":
        raise SystemExit("usage: prove_mgpu.py MNT4753|MNT6753 compute <params> <input> <output> [--unfused-c]")
    curve = {"MNT4753": 0, "MNT6753": 1}[sys.argv[1]]
    params_path, input_path, output_path = sys.argv[3:6]
    fused = "--unfused-c" not in sys.argv[6:]
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    share = os.environ.get("PROVE_SHARE_GPU") == "1"

    from __graft_entry__ import load_package
    pkg = load_package()
    dist = None
    device = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if share:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            device = torch.device("cuda", local_rank)
            dist.init_process_group("nccl", device_id=device)
    dev_index = 0 if share else local_rank
    pkg.init(dev_index)

    g1w, g2w = pkg.affine_words(curve, 1), pkg.affine_words(curve, 2)
    t0 = time.perf_counter()
    d, m = (int(v) for v in np.fromfile(params_path, dtype=np.uint64, count=2))
    # file layout: d, m, A[m+1] G1, B1[m+1] G1, B2[m+1] G2, L[m-1] G1, H[d] G1   (generate_parameters.cpp:60-85)
    layout = [("A", 1, g1w, m + 1), ("B1", 1, g1w, m + 1), ("B2", 2, g2w, m + 1), ("L", 1, g1w, m - 1), ("H", 1, g1w, d)]
    sets, spans, rows, off = {}, {}, {}, 16
    for name, group, words, n in layout:
        lo, hi = pkg.parallel.shard_range(n, rank, world)
        rows[name] = read_slice(params_path, off + lo * words * 8, hi - lo, words)
        spans[name] = (lo, hi)
        off += n * words * 8
    if fused:
        rows["C"] = np.concatenate([rows.pop("H"), rows.pop("L"), rows.pop("B1")])
        results = [("A", 1), ("B2", 2), ("C", 1)]
    else:
        results = [(name, group) for name, group, _, _ in layout]
    for name, group in results:
        sets[name] = pkg.BaseSet(curve, group, rows.pop(name))
    dom = pkg.Domain(curve, d + 1)   # twiddle tables depend on the parameters only (like the window tables)
    if world > 1:
        dist.barrier()
    t_params = time.perf_counter()

    # ---- timed window: input load + compute + output write (main.cpp:203-270) ----
    # input file: w[m+1], ca[d+1], cb[d+1], cc[d+1], r   (generate_parameters.cpp:88-108).  Same data-driven order as
    # host/main.cpp: this rank's range of w is streamed to the device first (mnt753_load_file_to_device), the MSMs that only need w are
    # enqueued -- the G2 one first --, this rank's vector of compute_H arrives while they run.
    n_w, n_c = m + 1, d + 1
    # the range of w this rank's slices multiply: A, B1, B2 take w[lo .. hi), L takes w[2 + lo .. 2 + hi)
    w_first = min(spans["A"][0], spans["L"][0] + 2)
    w_count = max(spans["A"][1], spans["L"][1] + 2) - w_first
    d_w = pkg.DeviceBuffer.from_file(input_path, 96 * w_first, 96 * w_count)
    w_at = lambda i: d_w.ptr.value + 96 * (i - w_first)        # device address of w[i]
    lh, hh = spans["H"]
    scal = {"A": (w_at, 0), "B1": (w_at, 0), "B2": (w_at, 0), "L": (w_at, 2)}

    def start(name):
        at, shift = scal[name]
        lo, hi = spans[name]
        sets[name].msm_start(at(lo + shift), hi - lo)

    for name in ("B2", "A") if fused else ("B2", "A", "B1", "L"):
        start(name)
    r = np.fromfile(input_path, dtype=np.uint64, count=12, offset=96 * n_w + 3 * 96 * n_c)
    # compute_H: who holds which vector (parallel.h_vector_home)
    home = pkg.parallel.h_vector_home(world)
    mine = [k for k in ("ca", "cb", "cc") if home[k] == rank]
    if world == 1:
        bufs = {k: pkg.DeviceBuffer.from_file(input_path, 96 * n_w + i * 96 * n_c, 96 * n_c) for i, k in enumerate(("ca", "cb", "cc"))}
        d_h = pkg.DeviceBuffer(96 * (d + 2))
        t_in = time.perf_counter()
        dom.compute_h(bufs["ca"].ptr.value, bufs["cb"].ptr.value, bufs["cc"].ptr.value, d_h.ptr.value)
        h_at = lambda i: d_h.ptr.value + 96 * i
    else:
        import torch
        tdev = torch.device("cuda", dev_index)
        vec = {k: torch.empty(12 * n_c, dtype=torch.int64, device=tdev) for k in (("ca", "cb", "cc") if rank == 0 else mine)}
        for i, k in enumerate(("ca", "cb", "cc")):
            if k in mine:
                pkg.api._check(pkg.lib().mnt753_load_file_to_device(input_path.encode(), 96 * n_w + i * 96 * n_c, 96 * n_c, vec[k].data_ptr()), "mnt753_load_file_to_device")
        t_in = time.perf_counter()
        for k in mine:
            dom.compute_h_chain(vec[k].data_ptr())
        pkg.lib().mnt753_sync(None)          # the chains ran on the library's default stream; the exchange uses torch.distributed's
        pkg.parallel.gather_chained_to_rank0(dist, rank, world, vec, via_host=share)
        h_mine = torch.empty(12 * max(hh - lh, 1), dtype=torch.int64, device=tdev)
        t_h = None
        if rank == 0:
            t_h = torch.empty(12 * (d + 2), dtype=torch.int64, device=tdev)
            # only the stream that carries the dependency: recv orders the arrived cb / cc against torch's current stream, which is
            # the null stream the library's transforms run on.  A device-wide synchronisation here would also wait for the w-MSMs
            # (B2, A) on their own non-blocking streams and serialise every rank's C behind them.
            torch.cuda.current_stream(tdev).synchronize()
            dom.compute_h_finish(vec["ca"].data_ptr(), vec["cb"].data_ptr(), vec["cc"].data_ptr(), t_h.data_ptr())
            pkg.lib().mnt753_sync(None)
        pkg.parallel.scatter_h_slices(dist, rank, world, d, t_h, h_mine, via_host=share)
        torch.cuda.current_stream(tdev).synchronize()    # the slice has arrived; the MSMs in flight are not waited for
        h_at = lambda i: h_mine.data_ptr() + 96 * (i - lh)
    if fused:
        # scalars of this rank's slice of the concatenated sum: h[lo_H, hi_H) | w[2 + lo_L, 2 + hi_L) | r * w[lo_B1, hi_B1)
        (ll, hl), (lb, hb) = spans["L"], spans["B1"]
        n_c_set = (hh - lh) + (hl - ll) + (hb - lb)
        d_sc = pkg.DeviceBuffer(96 * max(n_c_set, 1))
        pkg.copy_d2d(d_sc.ptr.value, h_at(lh), 96 * (hh - lh))
        pkg.copy_d2d(d_sc.ptr.value + 96 * (hh - lh), w_at(ll + 2), 96 * (hl - ll))
        pkg.vec_scale(curve, d_sc.ptr.value + 96 * ((hh - lh) + (hl - ll)), w_at(lb), r, hb - lb)
        sets["C"].msm_start(d_sc.ptr.value, n_c_set)
    else:
        sets["H"].msm_start(h_at(lh), hh - lh)
    partial = {name: sets[name].msm_finish() for name, _ in results}
    t_msm = time.perf_counter()

    # one exchange per proof: the partial points of this rank
    flat = np.concatenate([partial[name] for name, _ in results])
    if world > 1:
        gathered = pkg.parallel.all_gather_points(flat, device)
    else:
        gathered = [flat]
    total, pos = {}, 0
    for name, group in results:
        pw = pkg.projective_words(curve, group)
        total[name] = pkg.parallel.fold_partials(pkg.api, curve, group, [g[pos:pos + pw] for g in gathered])
        pos += pw
    t_fold = time.perf_counter()

    if rank == 0:
        if fused:
            c = total["C"]
        else:
            scaled = pkg.point_scale(curve, 1, r, total["B1"])
            c = pkg.point_add(curve, 1, total["H"], pkg.point_add(curve, 1, total["L"], scaled))
        with open(output_path, "wb") as f:
            pkg.point_to_affine(curve, 1, total["A"]).tofile(f)
            pkg.point_to_affine(curve, 2, total["B2"]).tofile(f)
            pkg.point_to_affine(curve, 1, c).tofile(f)
        t_out = time.perf_counter()
        print(json.dumps({"curve": sys.argv[1], "n_gpus": world, "d": d, "m": m,
                          "fused_c": fused, "load_params_s": t_params - t0, "input_streamed_behind_the_w_msms_s": t_in - t_params,
                          "compute_h_and_remaining_msm_s": t_msm - t_in, "exchange_and_fold_s": t_fold - t_msm,
                          "total_input_to_output_s": t_out - t_params}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
