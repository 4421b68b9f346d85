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


def test_shard_range_covers_everything():
    from __graft_entry__ import load_package
    pkg = load_package()
    for n in (0, 1, 7, 8, 1000, (1 << 20) + 1):
        for world in (1, 2, 3, 4, 8):
            spans = [pkg.parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
