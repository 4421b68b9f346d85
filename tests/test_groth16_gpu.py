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
    if not os.access(REF, os.X_OK):
        pytest.skip("oracle/_ref/ref_groth16 not built (needs the reference tree at build time)")
    v = subprocess.run([REF, "verify", NAME[curve], os.path.dirname(fx(curve, "vk.txt")), full], capture_output=True, text=True)
    assert v.returncode == 0 and "VERIFIED" in v.stdout, v.stdout + v.stderr
    # a wrong witness file size is refused before anything is allocated
    short = tmp_path / "short.bin"; short.write_bytes(open(fx(curve, "witness.bin"), "rb").read()[:-96])
    r = subprocess.run([EXE, NAME[curve], "compute-r1cs", fx(curve, "params.bin"), fx(curve, "r1cs.bin"), str(short), chal], capture_output=True, text=True)
    assert r.returncode == 1 and "witness file size" in r.stderr
