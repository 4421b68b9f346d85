"""Readers for tests/golden/*.bin (layouts documented in oracle/mint_golden.cpp)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CURVE_TAG = {0: "mnt4", 1: "mnt6"}


def aff_words(curve, group):
    return 24 * (1 if group == 1 else (2 if curve == 0 else 3))


def _load(name):
    return np.fromfile(os.path.join(GOLDEN, name), dtype=np.uint64)


def field(tag):
    """-> array [24, 8, 12]: a, b, a*b, a+b, a-b, a^-1, -a, as_bigint(a)"""
    return _load(f"field_{tag}.bin").reshape(24, 8, 12)


def group(curve, grp):
    aw = aff_words(curve, grp)
    raw = _load(f"group_{CURVE_TAG[curve]}_g{grp}.bin").reshape(8, 6 * aw + 12)
    out = []
    for rec in raw:
        P, Q, s = rec[:aw], rec[aw:2 * aw], rec[2 * aw:2 * aw + 12]
        rest = rec[2 * aw + 12:].reshape(4, aw)
        out.append(dict(P=P, Q=Q, s=s, sum=rest[0], dbl=rest[1], diff=rest[2], mul=rest[3]))
    return out


def extfield(curve):
    """-> array [24, 7, DEG * 12]: a, b, a*b, a^2, a^-1, a+b, a-b in Fq2 (MNT4753) / Fq3 (MNT6753)"""
    w = 12 * (2 if curve == 0 else 3)
    return _load(f"extfield_{CURVE_TAG[curve]}.bin").reshape(24, 7, w)


def groupkat(curve, grp):
    """-> list of dicts P, Q, sum (P+Q), dbl (2P), dbl_madd (2P + Q, mixed_add), dbl_add3 (2P + 3Q), diff (P-Q); all affine"""
    aw = aff_words(curve, grp)
    raw = _load(f"groupkat_{CURVE_TAG[curve]}_g{grp}.bin").reshape(16, 7, aw)
    keys = ("P", "Q", "sum", "dbl", "dbl_madd", "dbl_add3", "diff")
    return [dict(zip(keys, rec)) for rec in raw]


def msm(curve, grp, n):
    aw = aff_words(curve, grp)
    raw = _load(f"msm_{CURVE_TAG[curve]}_g{grp}_{n}.bin")
    bases = raw[:n * aw].reshape(n, aw)
    scalars = raw[n * aw:n * aw + 12 * n].reshape(n, 12)
    result = raw[n * aw + 12 * n:]
    assert result.size == aw
    return bases, scalars, result


MSM_SIZES = {1: [1, 2, 3, 17, 256, 1000], 2: [1, 2, 17, 128]}
FFT_LOGM = [1, 2, 3, 6, 10]
H_LOGM = [3, 8]


def fft(curve, logm):
    m = 1 << logm
    raw = _load(f"fft_{CURVE_TAG[curve]}_{logm}.bin").reshape(5, m, 12)
    return raw[0], raw[1:]  # input, [FFT, iFFT, cosetFFT, icosetFFT]


def h(curve, logm):
    m = 1 << logm
    raw = _load(f"h_{CURVE_TAG[curve]}_{logm}.bin")
    ca, cb, cc = (raw[i * 12 * m:(i + 1) * 12 * m].reshape(m, 12) for i in range(3))
    return ca, cb, cc, raw[3 * 12 * m:].reshape(m + 1, 12)


def e2e_fast_mnt6_paths():
    """the reference generator's own `fast` size for MNT6753 (generate_parameters.cpp:127-133, log2_d = 10) and the proof the
    reference's ./main wrote for it"""
    return tuple(os.path.join(GOLDEN, f"e2e_mnt6_2p10_{k}.bin") for k in ("params", "input", "output"))


def e2e_paths(curve):
    t = CURVE_TAG[curve]
    return tuple(os.path.join(GOLDEN, f"e2e_{t}_{k}.bin") for k in ("params", "input", "output"))
