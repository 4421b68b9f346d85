"""CPU: the oracle (oracle/mnt753_oracle.c) against vectors minted by the REFERENCE's own code
(oracle/mint_golden.cpp, tools/mint_e2e.sh).  This is what pins the oracle."""
import filecmp
import hashlib
import os

import numpy as np
import pytest

import golden_io as G
import oracle_lib as O


@pytest.mark.parametrize("tag,mod", [("A", 0), ("B", 1)])
def test_field_ops(tag, mod):
    for a, b, ab, s, d, inv, neg, big in G.field(tag):
        assert np.array_equal(O.field_op(mod, 0, a, b), ab)
        assert np.array_equal(O.field_op(mod, 1, a, b), s)
        assert np.array_equal(O.field_op(mod, 2, a, b), d)
        assert np.array_equal(O.field_op(mod, 3, a), inv)
        assert np.array_equal(O.field_op(mod, 5, a), neg)
        assert np.array_equal(O.field_op(mod, 4, a), big)


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("group", [1, 2])
def test_group_ops(curve, group):
    for r in G.group(curve, group):
        assert np.array_equal(O.point_op(curve, group, 0, r["P"], r["Q"]), r["sum"])
        assert np.array_equal(O.point_op(curve, group, 1, r["P"]), r["dbl"])
        assert np.array_equal(O.point_op(curve, group, 2, r["P"], r["Q"]), r["diff"])
        assert np.array_equal(O.point_op(curve, group, 3, r["P"], r["s"]), r["mul"])


@pytest.mark.parametrize("curve", [0, 1])
def test_extension_field_ops(curve):
    """Fq2 / Fq3 of the oracle against libff's Fp2_model / Fp3_model (extfield_<curve>.bin, second capture of oracle/mint_golden.cpp)."""
    for a, b, ab, sq, inv, s, d in G.extfield(curve):
        assert np.array_equal(O.ext_op(curve, 0, a, b), ab)
        assert np.array_equal(O.ext_op(curve, 1, a), sq)
        assert np.array_equal(O.ext_op(curve, 2, a), inv)
        assert np.array_equal(O.ext_op(curve, 3, a, b), s)
        assert np.array_equal(O.ext_op(curve, 4, a, b), d)


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("group", [1, 2])
def test_group_law_kats(curve, group):
    """operator+ / dbl / mixed_add of the reference's group classes: 16 cases per group with every side path (equal points, opposite
    points, identities on either side, mixed addition that meets an equal / opposite point)."""
    for r in G.groupkat(curve, group):
        P2 = O.point_op(curve, group, 1, r["P"])
        assert np.array_equal(O.point_op(curve, group, 0, r["P"], r["Q"]), r["sum"])
        assert np.array_equal(P2, r["dbl"])
        assert np.array_equal(O.point_op(curve, group, 0, P2, r["Q"]), r["dbl_madd"])
        Q3 = O.point_op(curve, group, 0, O.point_op(curve, group, 1, r["Q"]), r["Q"])
        assert np.array_equal(O.point_op(curve, group, 0, P2, Q3), r["dbl_add3"])
        assert np.array_equal(O.point_op(curve, group, 2, r["P"], r["Q"]), r["diff"])


def test_reference_generator_fast_set_mnt6753(tmp_path):
    """The reference generator's own `fast` size for MNT6753 (generate_parameters.cpp:127-133: d + 1 = 2^10, 1.77 MB of files with the
    generator's real R1CS-chain witness) and the proof the reference's ./main wrote for it."""
    params, inp, expected = G.e2e_fast_mnt6_paths()
    out = str(tmp_path / "proof.bin")
    O.prove(1, params, inp, out)
    assert filecmp.cmp(out, expected, shallow=False)


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("group,n", [(1, n) for n in G.MSM_SIZES[1]] + [(2, n) for n in G.MSM_SIZES[2]])
def test_msm(curve, group, n):
    bases, scalars, result = G.msm(curve, group, n)
    for chunks in (1, 4):   # the reference result does not depend on the OpenMP chunking
        assert np.array_equal(O.msm(curve, group, bases, scalars, chunks), result)


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("logm", G.FFT_LOGM)
def test_fft(curve, logm):
    v, outs = G.fft(curve, logm)
    for kind in range(4):
        assert np.array_equal(O.fft(curve, kind, v).reshape(-1, 12), outs[kind]), f"kind {kind}"


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("logm", G.H_LOGM)
def test_compute_h(curve, logm):
    ca, cb, cc, h = G.h(curve, logm)
    assert np.array_equal(O.compute_h(curve, ca, cb, cc).reshape(-1, 12), h)


@pytest.mark.parametrize("curve", [0, 1])
def test_end_to_end_proof(curve, tmp_path):
    params, inp, expected = G.e2e_paths(curve)
    out = str(tmp_path / "proof.bin")
    O.prove(curve, params, inp, out, chunks=2)
    assert filecmp.cmp(out, expected, shallow=False)


def test_golden_manifest():
    sums = os.path.join(G.GOLDEN, "SHA256SUMS")
    for line in open(sums):
        digest, name = line.split()
        assert hashlib.sha256(open(os.path.join(G.GOLDEN, name), "rb").read()).hexdigest() == digest, name


def test_reference_build_agrees_when_present(tmp_path):
    """Where oracle/_ref exists (built from /root/reference), the reference binary itself re-proves the e2e set."""
    ref_main = os.path.join(O.ROOT, "oracle", "_ref", "main")
    if not os.path.exists(ref_main):
        pytest.skip("oracle/_ref not built on this machine")
    import subprocess
    params, inp, expected = G.e2e_paths(1)
    out = str(tmp_path / "ref.bin")
    subprocess.check_call([ref_main, "MNT6753", "compute", params, inp, out], stdout=subprocess.DEVNULL)
    assert filecmp.cmp(out, expected, shallow=False)


@pytest.mark.parametrize("curve,group,n", [(0, 1, 256), (0, 2, 128), (1, 1, 256), (1, 2, 128)])
def test_libff_msm_driver_on_the_golden_records(curve, group, n, tmp_path):
    """oracle/_ref/ref_msm_bench (OUR driver around libff's multi_exp_with_mixed_addition<BDLO12>, the checker of the MSMs at size in
    tests/test_msm_gpu.py and bench.py's cpu_baseline) for all four (curve, group) pairs: its file format and its result words against
    the records the reference minted."""
    ref = os.path.join(O.ROOT, "oracle", "_ref", "ref_msm_bench")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref not built on this machine")
    import json
    import subprocess
    bases, scalars, result = G.msm(curve, group, n)
    path = tmp_path / "pairs.bin"
    with open(path, "wb") as f:
        bases.tofile(f); scalars.tofile(f)
    r = subprocess.run([ref, str(path), str(n), ("MNT4753", "MNT6753")[curve], f"G{group}"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1000:]
    hx = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["result_affine_hex"]
    got = np.array([int(hx[16 * i:16 * i + 16], 16) for i in range(len(hx) // 16)], dtype=np.uint64)
    assert np.array_equal(got, result)
