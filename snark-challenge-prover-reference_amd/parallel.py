"""Multi-GPU MSM: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The reference shards an MSM across OpenMP threads as contiguous slices and sums the partial results serially
(depends/libff/libff/algebra/scalar_multiplication/multiexp.tcc:402-441).  The same decomposition is used
across GPUs: rank g owns bases[lo_g:hi_g] resident in its HBM, runs the whole Pippenger on its slice, and the
only exchange is ONE projective point per rank (288 B for G1) -- an all_gather of 36..108 uint64 words, which is
latency-bound; the xGMI link bandwidth is irrelevant to it.  Elliptic-curve addition is not an RCCL reduction
operator, so the fold after the gather is W-1 host point additions (microseconds).
"""
import numpy as np


def shard_range(n, rank, world):
    """Contiguous slice of rank `rank`: one = n // world, the last rank takes the remainder (multiexp.tcc:417-431)."""
    one = n // world
    lo = rank * one
    hi = n if rank == world - 1 else (rank + 1) * one
    return lo, hi


def fold_partials(api, curve, group, partials):
    """Sum projective partial results in rank order (the reference's serial `final = final + partial[i]`)."""
    acc = np.ascontiguousarray(partials[0], dtype=np.uint64)
    for p in partials[1:]:
        acc = api.point_add(curve, group, acc, np.ascontiguousarray(p, dtype=np.uint64))
    return acc


def all_gather_points(local_words, device=None):
    """all_gather one projective point per rank.  local_words: numpy uint64 [W].  Returns list of numpy arrays."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    # uint64 has no collective support in torch; the words travel as int64 bit patterns
    t = torch.from_numpy(np.ascontiguousarray(local_words, dtype=np.uint64).view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [o.cpu().numpy().view(np.uint64) for o in out]


def msm_sharded(api, curve, group, local_partial, device=None):
    """local_partial: this rank's projective MSM result over its slice.  Returns the global sum on every rank."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return np.ascontiguousarray(local_partial, dtype=np.uint64)
    parts = all_gather_points(local_partial, device)
    return fold_partials(api, curve, group, parts)
