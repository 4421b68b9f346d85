"""CPU, world_size 2 and 4 over gloo: the multi-GPU decomposition of an MSM (contiguous slices, one projective point per
rank exchanged by all_gather, serial fold) -- with the oracle standing in for the per-rank device MSM, which needs
a GPU.  The exchange and fold code is the product's (snark-challenge-prover-reference_amd/parallel.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, curve, group, n, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from __graft_entry__ import load_package
    import oracle_lib as O
    pkg = load_package()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    pts = pkg.synth_points(curve, group, 5, n, threads=2)
    sc = pkg.synth_scalars(curve, 6, n)
    lo, hi = pkg.parallel.shard_range(n, rank, world)
    local_aff = O.msm(curve, group, pts[lo:hi], sc[lo:hi])            # stand-in for BaseSet.msm on this rank's GPU
    local = pkg.point_from_affine(curve, group, local_aff)
    total = pkg.parallel.msm_sharded(pkg.api, curve, group, local)
    again = pkg.parallel.msm_sharded(pkg.api, curve, group, local)     # the persistent exchange buffers are reusable
    assert np.array_equal(total, again)
    q.put((rank, pkg.point_to_affine(curve, group, total).tobytes(), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("curve,group,n,world", [(0, 1, 37, 2), (1, 2, 11, 2), (0, 1, 41, 4)])
def test_sharded_msm_over_gloo(curve, group, n, world):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from __graft_entry__ import load_package
    import oracle_lib as O
    pkg = load_package()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, curve, group, n, q)) for r in range(world)]
    for p in procs: p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    pts = pkg.synth_points(curve, group, 5, n, threads=2)
    sc = pkg.synth_scalars(curve, 6, n)
    expect = O.msm(curve, group, pts, sc).tobytes()
    assert sorted(r[2] for r in res) == [pkg.parallel.shard_range(n, r, world) for r in range(world)]
    assert all(r[1] == expect for r in res)


def _h_worker(rank, world, port, curve, logm, q):
    """compute_H spread over the ranks exactly as prove_mgpu.py does it -- parallel.h_vector_home / gather_chained_to_rank0 /
    scatter_h_slices are the product's -- with the oracle standing in for the device transforms (which need a GPU)."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    import oracle_lib as O
    pkg = load_package()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    m = 1 << logm
    d = m - 1
    home = pkg.parallel.h_vector_home(world)
    mine = [k for k in ("ca", "cb", "cc") if home[k] == rank]
    inputs = {k: pkg.synth_scalars(curve, 20 + i, m) for i, k in enumerate(("ca", "cb", "cc"))}
    as_t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64).reshape(-1).copy())
    vec = {k: torch.zeros(12 * m, dtype=torch.int64) for k in (("ca", "cb", "cc") if rank == 0 else mine)}
    for k in mine:   # x <- cosetFFT(iFFT(x)): mnt753_compute_h_chain on the device
        vec[k].copy_(as_t(O.fft(curve, 2, O.fft(curve, 1, inputs[k]))))
    pkg.parallel.gather_chained_to_rank0(dist, rank, world, vec)
    lo, hi = pkg.parallel.shard_range(d, rank, world)
    h_mine = torch.zeros(12 * max(hi - lo, 1), dtype=torch.int64)
    t_h = None
    if rank == 0:    # mnt753_compute_h_finish: a <- icosetFFT((a b - c) / Z), h <- a | 0
        mod = curve   # Fr of MNT4753 is modulus A (0), of MNT6753 modulus B (1)
        a, b, c = (vec[k].numpy().view(np.uint64).reshape(m, 12) for k in ("ca", "cb", "cc"))
        t = np.stack([O.field_op(mod, 2, O.field_op(mod, 0, x, y), z) for x, y, z in zip(a, b, c)])
        t = O.fft(curve, 3, O.divide_by_z_on_coset(curve, t)).reshape(m, 12)
        t_h = as_t(np.concatenate([t, np.zeros((1, 12), dtype=np.uint64)]))
    pkg.parallel.scatter_h_slices(dist, rank, world, d, t_h, h_mine)
    q.put((rank, (lo, hi), h_mine.numpy().view(np.uint64)[:12 * (hi - lo)].tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("curve,logm,world", [(0, 4, 2), (1, 3, 3), (0, 3, 4)])
def test_compute_h_spread_over_ranks_over_gloo(curve, logm, world):
    """ranks 0 / 1 / 2 transform ca / cb / cc, the transformed vectors meet on rank 0, every rank ends up with its slice of
    coefficients_for_H: together they are the oracle's compute_H of the same inputs."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from __graft_entry__ import load_package
    import oracle_lib as O
    pkg = load_package()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_h_worker, args=(r, world, port, curve, logm, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    m = 1 << logm
    ca, cb, cc = (pkg.synth_scalars(curve, 20 + i, m) for i in range(3))
    want = O.compute_h(curve, ca, cb, cc).reshape(m + 1, 12)
    got = b"".join(r[2] for r in res)
    assert [r[1] for r in res] == [pkg.parallel.shard_range(m - 1, r, world) for r in range(world)]
    assert got == want[:m - 1].tobytes()


def test_shard_range_covers_everything():
    from __graft_entry__ import load_package
    pkg = load_package()
    for n in (0, 1, 7, 8, 1000, (1 << 20) + 1):
        for world in (1, 2, 3, 4, 8):
            spans = [pkg.parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
