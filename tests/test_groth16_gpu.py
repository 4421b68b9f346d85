"""GPU: the witness-map front end on the device (n3) and the full proof loop (n4) -- SURVEY.md section 8f."""
import os
import subprocess

import numpy as np
import pytest

import golden_io as G
import oracle_lib as O
from test_groth16_cpu import EXE, NAME, REF, fx, load_input

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("curve", [0, 1])
def test_device_witness_evaluation_vs_reference(gpu, curve):
    d, m, w, ca, cb, cc, _ = load_input(curve)
    cs = gpu.R1cs.from_file(curve, fx(curve, "r1cs.bin"))
    assert cs.domain_size() == d + 1
    dw = gpu.DeviceBuffer.from_numpy(w)
    outs = [gpu.DeviceBuffer(96 * (d + 1)) for _ in range(3)]
    cs.evaluate(dw.ptr.value, outs[0].ptr.value, outs[1].ptr.value, outs[2].ptr.value, d + 1)
    for got, want in zip(outs, (ca, cb, cc)):
        assert np.array_equal(got.to_numpy().reshape(d + 1, 12), want)
    cs.close()


@pytest.mark.parametrize("curve", [0, 1])
def test_device_witness_evaluation_vs_oracle_random_system(gpu, curve):
    """A random sparse system much larger than the fixture: ragged rows (0 .. 9 terms), repeated variables, the constant column,
    zero coefficients, an output longer than nc + inputs + 1."""
    rng = np.random.default_rng(7 + curve)
    m, nc, num_inputs, out_len = 3000, 5000, 3, 8192
    w = gpu.synth_scalars(curve, 31, m + 1); w[0] = gpu.api.mont_one(curve)
    mats = []
    for k in range(3):
        counts = rng.integers(0, 10, size=nc); counts[0] = 0
        rp = np.zeros(nc + 1, dtype=np.uint64); rp[1:] = np.cumsum(counts)
        nnz = int(rp[nc])
        col = rng.integers(0, m + 1, size=nnz).astype(np.uint32)
        cf = gpu.synth_scalars(curve, 40 + k, nnz)
        cf[::17] = 0
        mats.append((rp, col, cf))
    cs = gpu.R1cs(curve, num_inputs, m, nc, mats)
    dw = gpu.DeviceBuffer.from_numpy(w)
    outs = [gpu.DeviceBuffer(96 * out_len) for _ in range(3)]
    cs.evaluate(dw.ptr.value, outs[0].ptr.value, outs[1].ptr.value, outs[2].ptr.value, out_len)
    want = O.r1cs_evaluate(curve, num_inputs, nc, mats, w, out_len)
    for got, exp in zip(outs, want):
        assert np.array_equal(got.to_numpy().reshape(out_len, 12), exp)
    with pytest.raises(gpu.Mnt753Error):
        cs.evaluate(dw.ptr.value, outs[0].ptr.value, outs[1].ptr.value, outs[2].ptr.value, nc)   # shorter than the system
    bad = [(mats[0][0], np.full_like(mats[0][1], m + 1), mats[0][2])] + mats[1:]
    with pytest.raises(gpu.Mnt753Error):
        gpu.R1cs(curve, num_inputs, m, nc, bad)                                                   # variable index out of range
    cs.close()


def synthetic_system(gpu, curve, nc, m, seed, pool=4096):
    """A ragged random constraint system at circuit scale: most rows have 1-4 terms, one in 64 has 40-200 (the shape of real gadget
    circuits); coefficients drawn from a pool of distinct field elements, every fifth term on the constant column."""
    rng = np.random.default_rng(seed)
    coeffs = gpu.synth_scalars(curve, 900 + seed, pool)
    mats = []
    for k in range(3):
        counts = rng.integers(1, 5, size=nc)
        heavy = rng.random(nc) < 1.0 / 64
        counts[heavy] = rng.integers(40, 200, size=int(heavy.sum()))
        counts[rng.random(nc) < 0.01] = 0
        rp = np.zeros(nc + 1, dtype=np.uint64); rp[1:] = np.cumsum(counts)
        nnz = int(rp[nc])
        col = rng.integers(0, m + 1, size=nnz).astype(np.uint32)
        col[::5] = 0
        cf = coeffs[rng.integers(0, pool, size=nnz)]
        mats.append((rp, col, cf))
    return mats


@pytest.mark.timeout(900)
def test_device_witness_evaluation_at_circuit_scale(gpu):
    """n3 at the size of BASELINE configs[3]: 2^20 - 8 constraints over 2^20 variables, ~10 M terms.  The device result is compared
    with the oracle (r1cs_to_qap.tcc:223-237 restated) on a sample of rows of every matrix -- random rows, the longest rows, the empty
    ones, the first and the last -- evaluated by the oracle as a sub-system over the same assignment; the input-consistency rows and
    the zero tail are checked in full."""
    curve, nc, m, num_inputs = 0, (1 << 20) - 8, (1 << 20) - 1, 5
    out_len = 1 << 20
    mats = synthetic_system(gpu, curve, nc, m, seed=11)
    w = gpu.synth_scalars(curve, 31, m + 1); w[0] = gpu.api.mont_one(curve)
    cs = gpu.R1cs(curve, num_inputs, m, nc, mats)
    dw = gpu.DeviceBuffer.from_numpy(w)
    outs = [gpu.DeviceBuffer(96 * out_len) for _ in range(3)]
    cs.evaluate(dw.ptr.value, outs[0].ptr.value, outs[1].ptr.value, outs[2].ptr.value, out_len)
    got = [o.to_numpy().reshape(out_len, 12) for o in outs]
    rng = np.random.default_rng(5)
    sub, rows_of = [], []
    for rp, col, cf in mats:
        lens = np.diff(rp.astype(np.int64))
        rows = np.unique(np.concatenate([rng.integers(0, nc, size=1500), np.argsort(lens)[-40:], np.flatnonzero(lens == 0)[:40], [0, nc - 1]]))
        srp = np.zeros(len(rows) + 1, dtype=np.uint64); srp[1:] = np.cumsum(lens[rows])
        idx = np.concatenate([np.arange(rp[r], rp[r + 1], dtype=np.int64) for r in rows]) if len(rows) else np.zeros(0, dtype=np.int64)
        sub.append((srp, col[idx].copy(), cf[idx].copy())); rows_of.append(rows)
    n_sub = max(len(r) for r in rows_of)
    for k in range(3):   # pad the shorter samples with empty rows so that the three matrices have one row count
        srp, c, f = sub[k]
        if len(srp) - 1 < n_sub:
            srp = np.concatenate([srp, np.full(n_sub - (len(srp) - 1), srp[-1], dtype=np.uint64)])
        sub[k] = (srp, c, f)
    want = O.r1cs_evaluate(curve, 0, n_sub, sub, w, n_sub + 1)
    for k in range(3):
        assert np.array_equal(got[k][rows_of[k]], want[k][:len(rows_of[k])]), f"matrix {k}"
    assert np.array_equal(got[0][nc:nc + num_inputs + 1], w[:num_inputs + 1])
    assert not got[0][nc + num_inputs + 1:].any() and not got[1][nc:].any() and not got[2][nc:].any()
    cs.close()


@pytest.mark.parametrize("curve", [0, 1])
def test_prove_from_constraint_system_and_verify(gpu, curve, tmp_path):
    """main_hip compute-r1cs (w and r only; ca / cb / cc from the constraint system on the device) writes the bytes the reference
    prover wrote from the full input file; completed by main_hip complete, the proof is accepted by the reference's verifier."""
    chal, full = str(tmp_path / "challenge.bin"), str(tmp_path / "full.bin")
    r = subprocess.run([EXE, NAME[curve], "compute-r1cs", fx(curve, "params.bin"), fx(curve, "r1cs.bin"), fx(curve, "witness.bin"), chal],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(chal, "rb").read() == open(fx(curve, "challenge.bin"), "rb").read()
    r = subprocess.run([EXE, NAME[curve], "compute", fx(curve, "params.bin"), fx(curve, "input.bin"), chal], capture_output=True, text=True)
    assert r.returncode == 0 and open(chal, "rb").read() == open(fx(curve, "challenge.bin"), "rb").read()
    r = subprocess.run([EXE, NAME[curve], "complete", fx(curve, "keys.bin"), fx(curve, "witness.bin"), chal, full, "--s-seed", "99"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    O.need_ref("ref_groth16")                     # missing = failure on a GPU box (tests/oracle_lib.py)
    v = subprocess.run([REF, "verify", NAME[curve], os.path.dirname(fx(curve, "vk.txt")), full], capture_output=True, text=True)
    assert v.returncode == 0 and "VERIFIED" in v.stdout, v.stdout + v.stderr
    # a wrong witness file size is refused before anything is allocated
    short = tmp_path / "short.bin"; short.write_bytes(open(fx(curve, "witness.bin"), "rb").read()[:-96])
    r = subprocess.run([EXE, NAME[curve], "compute-r1cs", fx(curve, "params.bin"), fx(curve, "r1cs.bin"), str(short), chal], capture_output=True, text=True)
    assert r.returncode == 1 and "witness file size" in r.stderr
