"""Multi-GPU MSM: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The reference shards an MSM across OpenMP threads as contiguous slices and sums the partial results serially
(depends/libff/libff/algebra/scalar_multiplication/multiexp.tcc:402-441).  The same decomposition is used
across GPUs: rank g owns bases[lo_g:hi_g] resident in its HBM, runs the whole Pippenger on its slice, and the
only exchange is ONE projective point per rank (288 B for G1) -- an all_gather of 36..108 uint64 words, which is
latency-bound; the xGMI link bandwidth is irrelevant to it.  Elliptic-curve addition is not an RCCL reduction
operator, so the fold after the gather is W-1 host point additions (microseconds).

The partial result of a rank arrives on the HOST (the last 20 doublings and additions of a bucket reduction are a
host Horner), so an exchange over RCCL is host -> device -> all_gather -> host.  `PointExchange` keeps
the four buffers of that round trip (pinned host in / out, device in / out) for the life of the process and uses
all_gather_into_tensor: no allocation, no tensor list, one synchronisation per exchange.
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


class PointExchange:
    """all_gather of `words` uint64 per rank with persistent buffers.  device = torch.device of this rank for RCCL, None for a
    CPU backend (gloo: the CPU tests and the shared-GPU development mode)."""

    def __init__(self, words, device=None):
        import torch
        import torch.distributed as dist
        self.words, self.device, self.world = int(words), device, dist.get_world_size()
        # uint64 has no collective support in torch; the words travel as int64 bit patterns
        self.h_in = torch.empty(self.words, dtype=torch.int64)
        self.h_out = torch.empty(self.world * self.words, dtype=torch.int64)
        if device is not None:
            self.h_in, self.h_out = self.h_in.pin_memory(), self.h_out.pin_memory()
            self.d_in = torch.empty(self.words, dtype=torch.int64, device=device)
            self.d_out = torch.empty(self.world * self.words, dtype=torch.int64, device=device)

    def all_gather(self, local_words):
        """local_words: numpy uint64 [words].  Returns the list of every rank's words (numpy uint64), in rank order."""
        import torch
        import torch.distributed as dist
        self.h_in.numpy()[:] = np.ascontiguousarray(local_words, dtype=np.uint64).view(np.int64)
        if self.device is not None:
            self.d_in.copy_(self.h_in, non_blocking=True)
            dist.all_gather_into_tensor(self.d_out, self.d_in)
            self.h_out.copy_(self.d_out, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
        else:
            dist.all_gather_into_tensor(self.h_out, self.h_in)
        out = self.h_out.numpy().view(np.uint64).reshape(self.world, self.words)
        return [out[r].copy() for r in range(self.world)]


_EXCHANGES = {}


def all_gather_points(local_words, device=None):
    """all_gather one block of words per rank (a projective point, or the five partial points of a proof)."""
    key = (len(local_words), str(device))
    ex = _EXCHANGES.get(key)
    if ex is None:
        ex = _EXCHANGES[key] = PointExchange(len(local_words), device)
    return ex.all_gather(local_words)


def msm_sharded(api, curve, group, local_partial, device=None):
    """local_partial: this rank's projective MSM result over its slice.  Returns the global sum on every rank."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return np.ascontiguousarray(local_partial, dtype=np.uint64)
    parts = all_gather_points(local_partial, device)
    return fold_partials(api, curve, group, parts)


# ---- compute_H over the ranks of a sharded prover ----------------------------------------------------------------------------------
# The FFT is not sharded, but ca, cb and cc are independent until the pointwise step (cuda_prover_piecewise.cu:24-34): ranks 0 / 1 / 2
# each stream ONE of them from the input file and run x <- cosetFFT(iFFT(x)) on it, the transformed cb and cc travel to rank 0
# (send / recv: RCCL over xGMI), rank 0 runs the pointwise step and the last transform and hands slice g of coefficients_for_H to
# rank g (the slice of H its MSM multiplies).  Two real exchange steps of the path, both point-to-point.
def h_vector_home(world):
    """rank that loads and transforms ca / cb / cc (rank 0 takes cc too when there are only two ranks)"""
    return {"ca": 0, "cb": 1 if world > 1 else 0, "cc": 2 if world > 2 else 0}


def _send(dist, t, dst, via_host):
    dist.send(t.cpu().contiguous() if via_host else t.contiguous(), dst)


def _recv(dist, t, src, via_host):
    if via_host:
        import torch
        tmp = torch.empty(t.shape, dtype=t.dtype)
        dist.recv(tmp, src)
        t.copy_(tmp)
    else:
        dist.recv(t, src)


def gather_chained_to_rank0(dist, rank, world, vec, via_host=False):
    """vec: dict name -> int64 tensor of 12 * (d + 1) words; rank 0 holds all three (its own chained, the others to be filled), ranks
    1 / 2 their own chained vector.  After the call rank 0 holds the three chained vectors."""
    home = h_vector_home(world)
    for k in ("cb", "cc"):
        if home[k] != 0:
            if rank == home[k]:
                _send(dist, vec[k], 0, via_host)
            elif rank == 0:
                _recv(dist, vec[k], home[k], via_host)


def scatter_h_slices(dist, rank, world, d, h_full, h_mine, via_host=False):
    """rank 0: h_full = coefficients_for_H (>= 12 * d words).  Every rank receives words of its slice [lo, hi) of d into h_mine."""
    lo, hi = shard_range(d, rank, world)
    if rank == 0:
        for g in range(1, world):
            glo, ghi = shard_range(d, g, world)
            if ghi > glo:
                _send(dist, h_full[12 * glo:12 * ghi], g, via_host)
        if hi > lo:
            h_mine[:12 * (hi - lo)].copy_(h_full[12 * lo:12 * hi])
    elif hi > lo:
        _recv(dist, h_mine[:12 * (hi - lo)], 0, via_host)
    return lo, hi
