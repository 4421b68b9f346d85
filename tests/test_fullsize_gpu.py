"""GPU: BASELINE.json's full-size configurations, pinned to the REFERENCE.

* configs[3] / configs[4]: the whole prove (main_hip, C++ host over the C ABI) at d = 2^20 - 1 (MNT4753) and d = 2^15 - 1
  (MNT6753) plus the `generate_parameters fast` sizes, on the seeded synthetic files of tools/synth_files.py; the sha256 of
  the proof must equal the one the reference prover (oracle/_ref/main = libsnark/main.cpp compiled from /root/reference) wrote
  for the same files in the build container -- tests/golden/oracle_hashes.json, minted by tools/mint_oracle_hashes.py
  ("Full-size parity without shipping big files").  The files themselves are checked against their
  recorded hashes first, so a platform-dependent generator cannot silently void the comparison.
* G2 MSM at the full sizes through the known discrete logs of the synthetic bases.
* compute_H at 2^16 and one 2^20 FFT element-for-element against the oracle.
* the device field layer directly against libff's golden field vectors (test hook mnt753_test_field_op)."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import golden_io as G
import oracle_lib as O
import synth_files

pytestmark = pytest.mark.gpu
EXE = os.path.join(O.ROOT, "snark-challenge-prover-reference_amd", "main_hip")
HASHES = json.load(open(os.path.join(G.GOLDEN, "oracle_hashes.json")))


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


@pytest.mark.parametrize("key", ["MNT6753_2p10", "MNT4753_2p14", "MNT6753_2p15", "MNT4753_2p20"])
@pytest.mark.timeout(1500)
def test_full_prove_matches_reference_hash(gpu, key, tmp_path):
    e = HASHES[key]
    curve = {"MNT4753": 0, "MNT6753": 1}[e["curve"]]
    params, inp, out = (str(tmp_path / k) for k in ("params", "input", "proof"))
    d, m = synth_files.write_files(gpu, curve, e["log2_d"], params, inp, seed=e["seed"])
    assert (d, m) == (e["d"], e["m"])
    assert os.path.getsize(params) == e["params_bytes"] and os.path.getsize(inp) == e["input_bytes"]
    assert sha256_file(params) == e["params_sha256"], "synthetic parameter file differs from the one the reference proved"
    assert sha256_file(inp) == e["input_sha256"], "synthetic input file differs from the one the reference proved"
    # [] = the CLI as the reference is invoked: a one-shot prover, no window tables (host/main.cpp); --tables = what a resident prover runs
    runs = [([], {}), (["--unfused-h", "--ref-order", "--tables"], {})]
    if key in ("MNT6753_2p15", "MNT4753_2p20"):
        # BASELINE configs[4] and the north-star split at its widest: every parameter vector cut into EIGHT contiguous slices
        # (multiexp.tcc:417-431; 4096 / 131072 points per slice), one base set, window table, input loader and stream per
        # logical device, H scattered by asynchronous peer copies, partial points folded in rank order.  The one-GPU test box
        # maps the eight logical devices onto its GPU (MNT753_SHARE_DEVICE=1): same code path, local instead of xGMI copies.
        runs.append((["--gpus", "8"], {"MNT753_SHARE_DEVICE": "1"}))
        runs.append((["--gpus", "4", "--ref-order", "--repeat", "2"], {"MNT753_SHARE_DEVICE": "1"}))
    for flags, env in runs:
        r = subprocess.run([EXE, e["curve"], "compute", params, inp, out] + flags, capture_output=True, text=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        assert os.path.getsize(out) == e["output_bytes"]
        assert sha256_file(out) == e["output_sha256"], f"proof differs from the reference's ({key}, flags {flags})"
        os.remove(out)
    if key == "MNT6753_2p15":
        # the process-per-GPU form of the same split (prove_mgpu.py, one all_gather of the five partial points per proof) at
        # world size 4; the ranks share the test GPU and exchange over gloo (PROVE_SHARE_GPU=1)
        import socket
        import sys
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(O.ROOT, "prove_mgpu.py"), e["curve"], "compute", params, inp, out]
        r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, PROVE_SHARE_GPU="1"), timeout=800)
        assert r.returncode == 0, r.stderr[-2000:]
        assert sha256_file(out) == e["output_sha256"], "prove_mgpu.py at world size 4 differs from the reference's proof"
        os.remove(out)
    os.remove(params); os.remove(inp)


@pytest.mark.parametrize("curve,logn", [(0, 20), (1, 15)])
def test_g2_msm_full_size_discrete_logs(gpu, curve, logn):
    """G2 MSM at BASELINE sizes: base[k] = e_k * G2_one, so the result is (sum s_k e_k mod r) * G2_one -- one host scalar
    multiplication that shares no code with the Pippenger kernels."""
    n = 1 << logn
    pts = gpu.synth_points(curve, 2, 77, n)
    sc = gpu.synth_scalars(curve, 78, n)
    pts[n - 1] = 0; pts[n - 2] = 0          # identity bases at the end, as in real B2 vectors
    sc[5] = 0; sc[6] = gpu.api.mont_one(curve)
    exp_sc = sc.copy(); exp_sc[n - 1] = 0; exp_sc[n - 2] = 0
    bs = gpu.BaseSet(curve, 2, pts)
    got = gpu.point_to_affine(curve, 2, bs.msm(sc))
    bs.close()
    assert np.array_equal(got, gpu.point_to_affine(curve, 2, gpu.synth_expected_msm(curve, 2, 77, exp_sc)))


@pytest.mark.parametrize("curve,logm", [(0, 16), (1, 15)])
def test_compute_h_large_vs_oracle(gpu, curve, logm):
    m = 1 << logm
    ca, cb, cc = (gpu.synth_scalars(curve, 500 + k, m) for k in range(3))
    dom = gpu.Domain(curve, m)
    a, b, c = (gpu.DeviceBuffer.from_numpy(x) for x in (ca, cb, cc))
    dh = gpu.DeviceBuffer(96 * (m + 1))
    dom.compute_h(a.ptr.value, b.ptr.value, c.ptr.value, dh.ptr.value)
    got = dh.to_numpy().reshape(m + 1, 12)
    dom.close()
    assert np.array_equal(got, O.compute_h(curve, ca, cb, cc).reshape(m + 1, 12))


def test_fft_2pow20_elementwise_vs_oracle(gpu):
    """BASELINE configs[2]: every one of the 2^20 outputs against the oracle's serial radix-2 FFT (restatement of
    basic_radix2_domain_aux.tcc:167-202), for the plain and the coset transform."""
    m = 1 << 20
    v = gpu.synth_scalars(0, 91, m)
    dom = gpu.Domain(0, m)
    for kind in (gpu.FFT, gpu.ICOSET_FFT):
        d = gpu.DeviceBuffer.from_numpy(v)
        dom.fft(kind, d.ptr.value)
        got = d.to_numpy()
        d.close()
        assert np.array_equal(got.reshape(m, 12), O.fft(0, kind, v).reshape(m, 12)), f"kind {kind}"
    dom.close()


@pytest.mark.parametrize("tag,mod", [("A", 0), ("B", 1)])
def test_device_field_layer_vs_libff_goldens(gpu, tag, mod):
    """fp_mul / fp_add / fp_sub / fp_inv / fp_neg / as_bigint on the device, element by element, against vectors minted
    by libff's Fp_model (tests/golden/field_*.bin), plus edge values 0, 1, p - 1 and cross-checks of the dedicated squaring,
    the fused two-product multiplier and the small-constant multiplier against compositions of the basic operations."""
    g = G.field(tag)
    a, b = g[:, 0].copy(), g[:, 1].copy()
    T = gpu.api.test_field_op
    assert np.array_equal(T(mod, 0, a, b).reshape(-1, 12), g[:, 2])
    assert np.array_equal(T(mod, 1, a, b).reshape(-1, 12), g[:, 3])
    assert np.array_equal(T(mod, 2, a, b).reshape(-1, 12), g[:, 4])
    assert np.array_equal(T(mod, 3, a).reshape(-1, 12), g[:, 5])
    assert np.array_equal(T(mod, 5, a).reshape(-1, 12), g[:, 6])
    assert np.array_equal(T(mod, 4, a).reshape(-1, 12), g[:, 7])
    assert np.array_equal(T(mod, 7, a).reshape(-1, 12), a)
    # edge values through the oracle (itself pinned to the same goldens): 0, 1, p - 1, 2, and the golden operands
    zero = np.zeros(12, dtype=np.uint64)
    one = g[0, 0].copy()                       # row 0 of the golden file has a = 1
    m1 = g[2, 0].copy()                        # row 2 has a = -1
    two = O.field_op(mod, 1, one, one)
    edge = np.stack([zero, one, m1, two] + [x for x in a[4:12]])
    for op in (3, 5, 4):
        want = np.stack([O.field_op(mod, op, x) for x in edge])
        assert np.array_equal(T(mod, op, edge).reshape(-1, 12), want), f"op {op}"
    eb = edge[::-1].copy()
    for op in (0, 1, 2):
        want = np.stack([O.field_op(mod, op, x, y) for x, y in zip(edge, eb)])
        assert np.array_equal(T(mod, op, edge, eb).reshape(-1, 12), want), f"op {op}"
    sq = np.stack([O.field_op(mod, 0, x, x) for x in edge])
    assert np.array_equal(T(mod, 6, edge).reshape(-1, 12), sq)
    fused = np.stack([O.field_op(mod, 1, O.field_op(mod, 0, x, y), O.field_op(mod, 0, x, x)) for x, y in zip(edge, eb)])
    assert np.array_equal(T(mod, 8, edge, eb).reshape(-1, 12), fused)
    thirteen = one.copy()
    for _ in range(12):
        thirteen = O.field_op(mod, 1, thirteen, one)
    small = np.stack([O.field_op(mod, 0, x, thirteen) for x in edge])
    assert np.array_equal(T(mod, 9, edge).reshape(-1, 12), small)
