"""CPU: the steps either side of the hot path against fixtures minted by the REFERENCE
(tools/mint_groth16.sh: generator, prover, completion and verifier of libsnark compiled from /root/reference):
  n3  evaluation of the constraint system on the assignment (r1cs_to_qap_witness_map, r1cs_to_qap.tcc:223-237) -- the oracle's
      restatement against the ca / cb / cc the reference's generator wrote;
  n4  completion of a challenge proof to a full Groth16 proof (main.cpp:312-319) -- the oracle's restatement AND the product's
      `main_hip complete` (host-only group operations through the C ABI) against the reference's completion, byte for byte;
      where the reference build is present, its verifier (r1cs_gg_ppzksnark_verifier_strong_IC) accepts the product's proof."""
import os
import subprocess

import numpy as np
import pytest

import golden_io as G
import oracle_lib as O

EXE = os.path.join(O.ROOT, "snark-challenge-prover-reference_amd", "main_hip")
REF = os.path.join(O.ROOT, "oracle", "_ref", "ref_groth16")
NAME = {0: "MNT4753", 1: "MNT6753"}


def fx(curve, name):
    return os.path.join(G.GOLDEN, f"g16_mnt{4 if curve == 0 else 6}", name)


def load_input(curve):
    """-> w (m + 1, 12), ca, cb, cc (d + 1, 12), r (12,) of the fixture's input.bin; d, m from params.bin"""
    d, m = (int(v) for v in np.fromfile(fx(curve, "params.bin"), dtype=np.uint64, count=2))
    raw = np.fromfile(fx(curve, "input.bin"), dtype=np.uint64).reshape(-1, 12)
    w, rest = raw[:m + 1], raw[m + 1:]
    return d, m, w, rest[:d + 1], rest[d + 1:2 * (d + 1)], rest[2 * (d + 1):3 * (d + 1)], rest[3 * (d + 1)]


@pytest.mark.parametrize("curve", [0, 1])
def test_oracle_witness_evaluation_vs_reference(pkg, curve):
    d, m, w, ca, cb, cc, _ = load_input(curve)
    num_inputs, m2, nc, mats = pkg.read_r1cs_file(fx(curve, "r1cs.bin"))
    assert m2 == m and nc + num_inputs + 1 == d + 1
    a, b, c = O.r1cs_evaluate(curve, num_inputs, nc, mats, w, d + 1)
    assert np.array_equal(a, ca) and np.array_equal(b, cb) and np.array_equal(c, cc)
    # the witness file is the input file without ca / cb / cc
    wit = np.fromfile(fx(curve, "witness.bin"), dtype=np.uint64).reshape(-1, 12)
    assert np.array_equal(wit[:-1], w) and wit.shape[0] == m + 2


@pytest.mark.parametrize("curve", [0, 1])
def test_oracle_proof_completion_vs_reference(curve):
    _, _, _, _, _, _, r = load_input(curve)
    keys = np.fromfile(fx(curve, "keys.bin"), dtype=np.uint64)
    chal = np.fromfile(fx(curve, "challenge.bin"), dtype=np.uint64)
    s = np.fromfile(fx(curve, "s.bin"), dtype=np.uint64)
    assert np.array_equal(O.complete_proof(curve, keys, chal, r, s), np.fromfile(fx(curve, "full.bin"), dtype=np.uint64))


@pytest.mark.parametrize("curve", [0, 1])
def test_product_proof_completion_vs_reference(curve, tmp_path):
    """main_hip complete: alpha / beta / delta / s terms added with the C ABI's host group operations (no GPU involved)."""
    out = str(tmp_path / "full.bin")
    for src in ("input.bin", "witness.bin"):       # r is the last element of either file
        r = subprocess.run([EXE, NAME[curve], "complete", fx(curve, "keys.bin"), fx(curve, src), fx(curve, "challenge.bin"), out,
                            "--s-file", fx(curve, "s.bin")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert open(out, "rb").read() == open(fx(curve, "full.bin"), "rb").read()
    # a different s gives a different, equally valid proof
    r = subprocess.run([EXE, NAME[curve], "complete", fx(curve, "keys.bin"), fx(curve, "input.bin"), fx(curve, "challenge.bin"), out, "--s-seed", "5"],
                       capture_output=True, text=True)
    assert r.returncode == 0 and open(out, "rb").read() != open(fx(curve, "full.bin"), "rb").read()
    if os.access(REF, os.X_OK):
        d = os.path.dirname(fx(curve, "vk.txt"))
        v = subprocess.run([REF, "verify", NAME[curve], d, out], capture_output=True, text=True)
        assert v.returncode == 0 and "VERIFIED" in v.stdout
        # the challenge proof itself (no alpha / beta / delta terms) and a tampered proof are rejected
        v = subprocess.run([REF, "verify", NAME[curve], d, fx(curve, "challenge.bin")], capture_output=True, text=True)
        assert v.returncode == 3 and "REJECTED" in v.stdout
        raw = bytearray(open(fx(curve, "full.bin"), "rb").read())
        other = open(out, "rb").read()
        g2b = len(raw) - 384
        raw[192:192 + g2b] = other[192:192 + g2b]  # B' (depends on s) of another proof with A', C' of this one
        bad = tmp_path / "bad.bin"; bad.write_bytes(raw)
        v = subprocess.run([REF, "verify", NAME[curve], d, str(bad)], capture_output=True, text=True)
        assert v.returncode == 3
