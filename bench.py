#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X Groth16 prover hot path (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

BASELINE.json's metric has two halves and the ONE JSON line rank 0 prints carries both:

  * `value` -- G1 MSM points/sec at 2^20 (MNT4753, configs[1]).  One process per GPU; a step is ONE multi-scalar
    multiplication over the rank's 2^20 (base, scalar) pairs, bases (with their window table) and scalars resident in
    HBM, followed -- when N > 1 -- by the only exchange the path has: an all_gather (RCCL) of one 288-byte projective point
    per rank and the serial fold of the N partial sums (multiexp.tcc:417-440).  Per-GPU work is fixed ("weak"); the same
    run also times the north-star split of ONE 2^20 array over the N ranks and reports it under `strong`.
  * `prove` -- Groth16 prove time in seconds, MNT4753, full-size parameters (configs[3]): `main_hip MNT4753 compute` (the
    C++ host over the C ABI) on the seeded synthetic files of tools/synth_files.py, timed by the prover itself with the
    reference's window ("Total time from input to output", libsnark/main.cpp:203-270) and by this script around the
    process; the sha256 of the proof is compared with the one the REFERENCE wrote for the same files
    (tests/golden/oracle_hashes.json, minted in the build container by tools/mint_oracle_hashes.py).  N = 1 only.

Secondary figures on the same line (N = 1): 2^20 FFT and compute_H (configs[2]), the G2 MSM at 2^20, the G1 MSM without
the window table and the one-off cost of building the table, `roofline` for the dominant kernels and `cpu_baseline`.

PyTorch is plumbing here (device memory for the scalars, streams/events, torch.distributed); every kernel is in
libmnt753_hip.so, reached through the C ABI.  oracle/ is used only for the cpu_baseline leg.
"""
import argparse
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_N = 20
ALGO_BYTES_PER_PAIR = 288          # 192 B affine G1 base + 96 B scalar, read once
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: 8 TB/s HBM3E
MODMUL_PEAK_PER_S = 22.0e9         # measured chip peak of the 753-bit Montgomery multiplier (profiles/r01/mulbench_mi355x.txt)
PROFILE_ROUND = "r02"


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
    return h.hexdigest()


def kernels_fingerprint():
    """sha256 over the MSM kernel sources: the PMC traffic figure in profiles/ is only quoted while it matches."""
    h = hashlib.sha256()
    for name in ("msm_kernels.hip.h", "msm_host.hpp", "curve753.hip.h", "fp753.hip.h"):
        with open(os.path.join(ROOT, "snark-challenge-prover-reference_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start the documented torchrun command as a CHILD before anything
    touches the GPU (never exec after HIP initialisation) and pass its exit code on."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# ---- CPU baseline ------------------------------------------------------------------------------------------
def cpu_baseline(pkg, pts, sc, np):
    """The reference's own multi_exp (BDLO12, chunks = host threads -- what B::multiexp_G1 runs) through oracle/_ref/ref_msm_bench
    when that build is present (kind "reference"), else the oracle's restatement of it (kind "port"), on a bounded prefix of
    the benchmark input.  Returns (dict, affine result, sample size)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_msm_bench")
    cores = os.cpu_count() or 1
    if os.access(ref, os.X_OK):
        # >= 2^17 points per chunk would need cores * 2^17 points; bound the sample at 2^19 (about 10-30 s on a large host)
        n2 = min(1 << 19, len(pts))
        with tempfile.NamedTemporaryFile(dir=os.environ.get("TMPDIR", "/tmp"), suffix=".bin") as f:
            pts[:n2].tofile(f); sc[:n2].tofile(f); f.flush()
            r = subprocess.run([ref, f.name, str(n2)], capture_output=True, text=True)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and lines:
            j = json.loads(lines[-1])
            got = np.array([int(j["result_affine_hex"][16 * i:16 * i + 16], 16) for i in range(24)], dtype=np.uint64)
            return dict(value=j["points_per_s"], unit="points/s", cores=j["threads"], kind="reference",
                        sample=f"first 2^{n2.bit_length() - 1} (base, scalar) pairs of the benchmark input; libff multi_exp_with_mixed_addition<BDLO12> of the "
                               f"reference compiled by oracle/build_ref.sh, {j['threads']} OpenMP chunks of {n2 // j['threads']} points, {j['seconds']:.1f} s"), got, n2
    import oracle_lib as O
    threads = O.lib().oracle_max_threads()
    n = 1 << 12
    t0 = time.time(); O.msm(0, 1, pts[:n], sc[:n], chunks=threads); t_small = time.time() - t0
    n2 = n
    while n2 < (1 << 19) and t_small * (n2 * 2 / n) < 15.0:   # scale the sample towards ~12 s of CPU work
        n2 *= 2
    t0 = time.time(); got = O.msm(0, 1, pts[:n2], sc[:n2], chunks=threads); dt = time.time() - t0
    return dict(value=n2 / dt, unit="points/s", cores=threads, kind="port",
                sample=f"first 2^{n2.bit_length() - 1} (base, scalar) pairs of the benchmark input, oracle chunked BDLO12 ({threads} chunks of {n2 // threads} points; "
                       f"measured 1.5x slower than the reference's GMP build on 8 cores, BASELINE.md), {dt:.1f} s"), got, n2


# ---- full prove ----------------------------------------------------------------------------------------------
def prove_leg(log2_d=20, curve_name="MNT4753"):
    """main_hip on the seeded synthetic files; hash compared with the reference-minted one.

    Runs in child processes BEFORE this process initialises the GPU: the metric is the reference's -- a fresh `./main` on
    an otherwise idle device.  A harness that already holds a GPU context and freed device memory measured the same prover
    4-8 % slower (0.202-0.213 s against 0.194-0.198 s stand-alone, profiles/r02) and its parameter load 25 % slower."""
    exe = os.path.join(ROOT, "snark-challenge-prover-reference_amd", "main_hip")
    key = f"{curve_name}_2p{log2_d}"
    expected = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_hashes.json"))).get(key)
    work = tempfile.mkdtemp(prefix="bench_prove_", dir=os.environ.get("TMPDIR", "/tmp"))
    pp, ip, op = (os.path.join(work, k) for k in ("params", "input", "output"))
    out = {"curve": curve_name, "log2_d": log2_d}
    try:
        t0 = time.time()
        g = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "synth_files.py"), curve_name, str(log2_d), pp, ip], capture_output=True, text=True)
        if g.returncode != 0:
            out.update(error=g.stderr[-400:], parity_ok=False)
            return out
        d = (1 << log2_d) - 1
        out.update(d=d, m=d + 1, synth_files_s=round(time.time() - t0, 2))
        files_ok = bool(expected) and sha256_file(pp) == expected["params_sha256"] and sha256_file(ip) == expected["input_sha256"]
        t0 = time.time()
        # three proofs of the same input in ONE process: the first is the reference's metric (fresh process, parameters loaded,
        # then input -> output); the others show what a resident prover pays per proof
        r = subprocess.run([exe, curve_name, "compute", pp, ip, op, "--repeat", "3"], capture_output=True, text=True)
        wall = time.time() - t0
        if r.returncode != 0:
            out.update(error=r.stderr[-400:], parity_ok=False)
            return out
        m1 = re.search(r"Total time from input to output: ([0-9.]+)s", r.stdout)   # the first proof
        m2 = re.search(r"load params: ([0-9.]+)s", r.stdout)
        sha = sha256_file(op)
        out["prover_stdout_first_proof"] = [l for l in r.stdout.strip().splitlines()][:9]
        out.update(input_to_output_s=float(m1.group(1)) if m1 else None, load_params_s=float(m2.group(1)) if m2 else None,
                   wall_incl_params_s=round(wall, 3), sha256=sha, sha256_expected=expected["output_sha256"] if expected else None,
                   synthetic_files_match_minted=files_ok, parity_ok=bool(expected) and files_ok and sha == expected["output_sha256"],
                   timing_window="libsnark/main.cpp:203-270 (input load + compute + output write; parameters resident)",
                   expected_from="tests/golden/oracle_hashes.json: oracle/_ref/main (the reference, bos_coster) on the same seeded files")
        # a second proof against resident parameters (main_hip batch mode): what a proving service pays per proof
        m3 = re.findall(r"Total time from input to output: ([0-9.]+)s", r.stdout)
        if len(m3) > 1:
            out["input_to_output_s_all"] = [float(x) for x in m3]
            out["resident_proof_s"] = min(float(x) for x in m3[1:])
    finally:
        for p in (pp, ip, op):
            if os.path.exists(p):
                os.remove(p)
        os.rmdir(work)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=LOG_N, help="log2 of the bases per GPU (benchmark config: 20)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prove", action="store_true", help="skip the full-prove leg (N = 1 runs it by default)")
    ap.add_argument("--no-extras", action="store_true", help="skip the FFT / compute_H / G2 / table-less legs")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(respawn_under_torchrun(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); run `python bench.py --gpus N` or torchrun with --nproc-per-node N")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    # the full-prove leg first, in child processes, while this process has not touched the GPU yet (see prove_leg)
    prove = None
    if world == 1 and not args.no_prove and args.log_n == LOG_N:
        prove = prove_leg()

    import numpy as np
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
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bases = pkg.BaseSet(0, 1, pts)                                  # parameters: resident before timing (main.cpp:201-203)
    torch.cuda.synchronize()
    precompute_ms = (time.perf_counter() - t0) * 1e3
    d_sc = torch.from_numpy(sc.view(np.int64)).to(device)          # scalars resident in HBM
    stream = torch.cuda.current_stream().cuda_stream
    comm_dev = None if share else device

    def step():
        local = bases.msm(d_sc.data_ptr(), n=n, on_device=True, stream=stream)
        return pkg.parallel.msm_sharded(pkg.api, 0, 1, local, comm_dev)

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            res = fn()
        phases = []
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = fn()
            phases.append(pkg.msm_last_timing())
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return res, elapsed, phases

    out, elapsed, tot_ms = timed(step, args.steps, args.warmup)
    acc_ms = [t["accumulate_ms"] for t in tot_ms]
    plan = pkg.msm_last_plan()

    # parity of what was just timed: every rank's slice through its discrete logs, folded like the timed path
    exp_local = pkg.synth_expected_msm(0, 1, 42 + 1000 * rank, sc)
    exp = pkg.parallel.msm_sharded(pkg.api, 0, 1, exp_local, comm_dev)
    ok = bool(np.array_equal(pkg.point_to_affine(0, 1, out), pkg.point_to_affine(0, 1, exp)))

    # strong scaling, the north-star split: ONE 2^log_n array (rank 0's), contiguous slice per rank (multiexp.tcc:417-431)
    strong = None
    if world > 1:
        lo, hi = pkg.parallel.shard_range(n, rank, world)
        pts0 = pts if rank == 0 else pkg.synth_points(0, 1, 42, n)
        sc0 = sc if rank == 0 else pkg.synth_scalars(0, 43, n)
        sl = pkg.BaseSet(0, 1, pts0[lo:hi])
        d_sl = torch.from_numpy(sc0[lo:hi].copy().view(np.int64)).to(device)

        def step_strong():
            local = sl.msm(d_sl.data_ptr(), n=hi - lo, on_device=True, stream=stream)
            return pkg.parallel.msm_sharded(pkg.api, 0, 1, local, comm_dev)

        s_out, s_elapsed, _ = timed(step_strong, args.steps, args.warmup)
        s_ok = bool(np.array_equal(pkg.point_to_affine(0, 1, s_out), pkg.point_to_affine(0, 1, pkg.synth_expected_msm(0, 1, 42, sc0))))
        strong = {"scaling": "strong", "workload": f"one 2^{args.log_n} MNT4753 G1 MSM split into {world} contiguous slices", "value": n * args.steps / s_elapsed,
                  "unit": "points/s", "ms_per_step": s_elapsed / args.steps * 1e3, "parity_ok": s_ok}
        ok = ok and s_ok
        sl.close()

    line = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * n * args.steps / elapsed
        acc = float(np.mean(acc_ms))
        achieved = ALGO_BYTES_PER_PAIR * n / (acc * 1e-3) / 1e9
        plan_c, windows = plan["window_bits"], plan["windows"]
        levels = plan.get("pair_levels", 0)
        # Montgomery products per sorted entry: an affine pair addition is 5 products + 1 squaring (0.76 of a product) including
        # the 3 of the simultaneous inversion, on 1/2, 1/4, ... of the entries; one divstep inversion (~94 products) per lane and
        # level over B = slots / 65536 lanes (at least 48); 11 per mixed addition on what is left
        slots = [windows * n / 2 ** l for l in range(1, levels + 1)]
        prod_per_entry = sum((5.76 + 94.0 / max(48.0, sl / 65536.0)) / 2 ** l for l, sl in zip(range(1, levels + 1), slots)) + 11.0 / 2 ** levels
        kernel_name = "k_bucket_accumulate<Mnt4G1>" if levels == 0 else f"k_pair_level<Mnt4G1> x{levels} + k_bucket_accumulate<Mnt4G1>"
        # PMC traffic of the same phase (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/collect_profiles.sh); quoted only
        # while the kernel sources are the ones it was measured on
        traffic, traffic_info = None, None
        tpath = os.path.join(ROOT, "profiles", PROFILE_ROUND, "accumulate_traffic.json")
        if os.path.exists(tpath) and args.log_n == LOG_N:
            tj = json.load(open(tpath))
            if tj.get("kernels_fingerprint") == kernels_fingerprint():
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_info = {"source": f"profiles/{PROFILE_ROUND}/accumulate_traffic.json", "kernels_fingerprint": tj.get("kernels_fingerprint"),
                                "achieved_GBps": traffic / (acc * 1e-3) / 1e9 if traffic else None,
                                "frac_of_hbm_peak": traffic / (acc * 1e-3) / 1e9 / HBM_PEAK_GBPS if traffic else None}
            else:
                traffic_info = {"stale": True, "note": "profiles traffic was measured on different kernel sources; re-run tools/collect_profiles.sh"}
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
                         "traffic": traffic, "traffic_info": traffic_info, "kernel": kernel_name, "kernel_ms": acc,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_PAIR * n,
                         "note": "bucket accumulation phase of one MSM (HIP events on the launch stream). `achieved` is ALGORITHMIC bytes / time as the "
                                 "contract asks; the phase is bound by the 753-bit multiplier (modmul_frac) and, in the first pairing level, by scattered "
                                 "table-row gathers (traffic_info.frac_of_hbm_peak), not by streaming",
                         "pair_levels": levels, "products_per_entry": prod_per_entry,
                         "modmul_per_s": prod_per_entry * windows * n / (acc * 1e-3), "modmul_peak_per_s": MODMUL_PEAK_PER_S,
                         "modmul_frac": prod_per_entry * windows * n / (acc * 1e-3) / MODMUL_PEAK_PER_S,
                         "mixed_addition_equivalents_per_s": windows * n / (acc * 1e-3)},
            "phases_ms": {k: float(np.mean([t[k] for t in tot_ms])) for k in tot_ms[0]},
            "precompute_ms": precompute_ms,
        }
        if strong:
            line["strong"] = strong

    if world == 1:
        if not args.no_cpu_baseline:
            base, got, n2 = cpu_baseline(pkg, pts, sc, np)
            chk = pkg.BaseSet(0, 1, pts[:n2])
            same = bool(np.array_equal(pkg.point_to_affine(0, 1, chk.msm(sc[:n2])), got))
            chk.close()
            base["matches_gpu_on_sample"] = same
            line["cpu_baseline"] = base
            ok = ok and same
        if not args.no_extras and args.log_n == LOG_N:
            extras = {}
            # G1 MSM without the window table (what a caller pays who cannot keep 8.9 GB per base set resident)
            os.environ["MNT753_MSM_PRECOMP"] = "0"
            nt = pkg.BaseSet(0, 1, pts)
            del os.environ["MNT753_MSM_PRECOMP"]
            _, e2, _ = timed(lambda: nt.msm(d_sc.data_ptr(), n=n, on_device=True, stream=stream), max(3, args.steps // 2), 1)
            extras["no_table_ms_per_step"] = e2 / max(3, args.steps // 2) * 1e3
            nt.close()
            bases.close()
            # 2^20 FFT and compute_H over Fr(MNT4753): BASELINE configs[2]; 192 B algorithmic per element per transform
            m = 1 << 20
            dom = pkg.Domain(0, m)
            vecs = [torch.from_numpy(pkg.synth_scalars(0, 70 + k, m).view(np.int64)).to(device) for k in range(3)]
            dh = torch.empty((m + 1) * 12, dtype=torch.int64, device=device)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

            def ev_time(fn, reps):
                fn(); torch.cuda.synchronize()
                ev0.record()
                for _ in range(reps):
                    fn()
                ev1.record(); torch.cuda.synchronize()
                return ev0.elapsed_time(ev1) / reps

            extras["fft_2p20_ms"] = ev_time(lambda: dom.fft(pkg.FFT, vecs[0].data_ptr(), stream=stream), 20)
            extras["fft_2p20_algorithmic_GBps"] = 192.0 * m / (extras["fft_2p20_ms"] * 1e-3) / 1e9
            extras["compute_h_2p20_ms"] = ev_time(lambda: dom.compute_h(vecs[0].data_ptr(), vecs[1].data_ptr(), vecs[2].data_ptr(), dh.data_ptr(), stream=stream), 5)
            dom.close(); del vecs, dh
            # G2 MSM at 2^20 (Fq2): 480 B algorithmic per pair
            g2 = pkg.synth_points(0, 2, 52, n)
            b2 = pkg.BaseSet(0, 2, g2)
            r2, e3, ph = timed(lambda: b2.msm(d_sc.data_ptr(), n=n, on_device=True, stream=stream), 3, 1)
            extras["g2_msm_2p20_ms"] = e3 / 3 * 1e3
            extras["g2_msm_points_per_s"] = n * 3 / e3
            extras["g2_parity_ok"] = bool(np.array_equal(pkg.point_to_affine(0, 2, r2), pkg.point_to_affine(0, 2, pkg.synth_expected_msm(0, 2, 52, sc))))
            ok = ok and extras["g2_parity_ok"]
            b2.close(); del g2
            line["extras"] = extras
        else:
            bases.close()
        del d_sc
        torch.cuda.empty_cache()
        if prove is not None:
            line["prove"] = prove
            ok = ok and bool(prove.get("parity_ok"))
        line["parity_ok"] = ok
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("bench.py: PARITY FAILURE")


if __name__ == "__main__":
    main()
