"""GPU: FFT / vector ops / compute_H parity -- HIP path vs the reference's golden vectors, the CPU oracle, and
size-independent properties at the full 2^20 domain.  Bit-exact."""
import numpy as np
import pytest

import golden_io as G
import oracle_lib as O

pytestmark = pytest.mark.gpu


def run_fft(pkg, curve, kind, v):
    m = v.shape[0]
    dom = pkg.Domain(curve, m)
    d = pkg.DeviceBuffer.from_numpy(v)
    dom.fft(kind, d.ptr.value)
    out = d.to_numpy().reshape(m, 12)
    dom.close(); d.close()
    return out


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("logm", G.FFT_LOGM)
def test_golden_fft(gpu, curve, logm):
    v, outs = G.fft(curve, logm)
    for kind in range(4):
        assert np.array_equal(run_fft(gpu, curve, kind, v), outs[kind]), f"kind {kind}"


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("logm", G.H_LOGM)
def test_golden_compute_h(gpu, curve, logm):
    ca, cb, cc, h = G.h(curve, logm)
    m = 1 << logm
    dom = gpu.Domain(curve, m)
    a, b, c = (gpu.DeviceBuffer.from_numpy(x) for x in (ca, cb, cc))
    dh = gpu.DeviceBuffer(96 * (m + 1))
    dom.compute_h(a.ptr.value, b.ptr.value, c.ptr.value, dh.ptr.value)
    assert np.array_equal(dh.to_numpy().reshape(m + 1, 12), h)
    # the unfused sequence of B:: calls (cuda_prover_piecewise.cu:24-47) gives the same vector
    a, b, c = (gpu.DeviceBuffer.from_numpy(x) for x in (ca, cb, cc))
    for v in (a, b):
        dom.fft(gpu.IFFT, v.ptr.value)
    for v in (a, b):
        dom.fft(gpu.COSET_FFT, v.ptr.value)
    gpu.vec_muleq(curve, a.ptr.value, b.ptr.value, m)
    dom.fft(gpu.IFFT, c.ptr.value); dom.fft(gpu.COSET_FFT, c.ptr.value)
    gpu.vec_subeq(curve, a.ptr.value, c.ptr.value, m)
    dom.divide_by_z_on_coset(a.ptr.value)
    dom.fft(gpu.ICOSET_FFT, a.ptr.value)
    assert np.array_equal(a.to_numpy().reshape(m, 12), h[:m])


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("logm", [4, 7, 9, 12, 13])
def test_vs_oracle_seeded(gpu, curve, logm):
    m = 1 << logm
    v = gpu.synth_scalars(curve, 300 + logm, m)
    for kind in range(4):
        assert np.array_equal(run_fft(gpu, curve, kind, v), O.fft(curve, kind, v).reshape(m, 12)), f"kind {kind}"


def test_domain_limits(gpu):
    for bad in (0, 1, 3, 12, 1000):
        with pytest.raises(gpu.Mnt753Error):
            gpu.Domain(0, bad)
    with pytest.raises(gpu.Mnt753Error):
        gpu.Domain(1, 1 << 16)       # MNT6753 Fr has two-adicity 15 (mnt6753_init.cpp:66)
    gpu.Domain(1, 1 << 15).close()   # ... and 2^15 is exactly the full-size MNT6753 domain


def test_vector_ops_vs_oracle(gpu):
    for curve in (0, 1):
        n = 777   # ragged: not a multiple of the block size
        a, b = gpu.synth_scalars(curve, 1, n), gpu.synth_scalars(curve, 2, n)
        da, db = gpu.DeviceBuffer.from_numpy(a), gpu.DeviceBuffer.from_numpy(b)
        gpu.vec_muleq(curve, da.ptr.value, db.ptr.value, n)
        prod = np.array([O.field_op(curve, 0, a[i], b[i]) for i in range(n)])
        assert np.array_equal(da.to_numpy().reshape(n, 12), prod)
        gpu.vec_subeq(curve, da.ptr.value, db.ptr.value, n)
        diff = np.array([O.field_op(curve, 2, prod[i], b[i]) for i in range(n)])
        assert np.array_equal(da.to_numpy().reshape(n, 12), diff)
        gpu.vec_muleq(curve, da.ptr.value, db.ptr.value, 0)   # empty is a no-op
        # dst = src * k with k on the host (the factor r of r * Bt1 folded into the scalars of B::groth16_C); in place as well
        k = gpu.synth_scalars(curve, 3, 1)[0]
        dd = gpu.DeviceBuffer.from_numpy(np.zeros_like(a))
        gpu.vec_scale(curve, dd.ptr.value, db.ptr.value, k, n)
        scaled = np.array([O.field_op(curve, 0, b[i], k) for i in range(n)])
        assert np.array_equal(dd.to_numpy().reshape(n, 12), scaled)
        gpu.vec_scale(curve, db.ptr.value, db.ptr.value, k, n)
        assert np.array_equal(db.to_numpy().reshape(n, 12), scaled)
        gpu.vec_scale(curve, db.ptr.value, db.ptr.value, k, 0)


def test_full_size_2pow20_properties(gpu):
    """BASELINE config[2]: domain 2^20 over Fr(MNT4753).  Round trips, linearity, and a spot check of
    FFT(v)[k] = sum_i v_i w^(ik) against the oracle on a sparse vector (size-independent, exact)."""
    m = 1 << 20
    dom = gpu.Domain(0, m)
    v = gpu.synth_scalars(0, 5, m)
    d = gpu.DeviceBuffer.from_numpy(v)
    dom.fft(gpu.FFT, d.ptr.value); dom.fft(gpu.IFFT, d.ptr.value)
    assert np.array_equal(d.to_numpy().reshape(m, 12), v)
    dom.fft(gpu.COSET_FFT, d.ptr.value); dom.fft(gpu.ICOSET_FFT, d.ptr.value)
    assert np.array_equal(d.to_numpy().reshape(m, 12), v)
    # linearity: FFT(a) - FFT(b) == FFT(a - b)
    w = gpu.synth_scalars(0, 6, m)
    da, db, dc = gpu.DeviceBuffer.from_numpy(v), gpu.DeviceBuffer.from_numpy(w), gpu.DeviceBuffer.from_numpy(v)
    gpu.vec_subeq(0, dc.ptr.value, db.ptr.value, m)
    for x in (da, db, dc):
        dom.fft(gpu.FFT, x.ptr.value)
    gpu.vec_subeq(0, da.ptr.value, db.ptr.value, m)
    assert np.array_equal(da.to_numpy(), dc.to_numpy())
    # sparse vector: only 3 non-zero entries -> every output is a 3-term sum the oracle evaluates directly:
    # FFT of e_j is the vector w^(jk); take a 2^10 oracle FFT of the decimated problem
    sp = np.zeros((m, 12), dtype=np.uint64)
    stride = m >> 10
    small = gpu.synth_scalars(0, 8, 1 << 10)
    sp[::stride] = small                      # v'_i = small_{i/stride}: FFT_m(v')[k] = FFT_1024(small)[k mod 1024]
    ds = gpu.DeviceBuffer.from_numpy(sp)
    dom.fft(gpu.FFT, ds.ptr.value)
    got = ds.to_numpy().reshape(m, 12)
    ref = O.fft(0, 0, small).reshape(1 << 10, 12)
    assert np.array_equal(got[:1 << 10], ref) and np.array_equal(got[5 << 10:6 << 10], ref)
    dom.close()
