"""ctypes binding of include/mnt753_hip.h.  No compute happens in Python; every call lands in
libmnt753_hip.so, and the loader raises if the library was not built (no fallback path)."""
import ctypes as C
import os

import numpy as np

CURVE_MNT4753, CURVE_MNT6753 = 0, 1
G1, G2 = 1, 2
FFT, IFFT, COSET_FFT, ICOSET_FFT = 0, 1, 2, 3

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Mnt753Error(RuntimeError):
    pass


def lib_path():
    """The product library next to this file; MNT753_LIB names another build of it (development A/B runs: an experimental
    variant is loaded from where it was built instead of being copied over the product file)."""
    return os.environ.get("MNT753_LIB") or os.path.join(_HERE, "libmnt753_hip.so")


def lib():
    """Load libmnt753_hip.so (built by `make` / __graft_entry__.build()); fail loudly if missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise Mnt753Error(f"{path} not found: build the HIP extension first (make, or __graft_entry__.build())")
    # RTLD_GLOBAL: the test library beside it (test_lib) names the product by its soname and must find THIS copy -- also when
    # MNT753_LIB loaded a development variant from another directory
    L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    u64p, vp, sz, i = C.POINTER(C.c_uint64), C.c_void_p, C.c_size_t, C.c_int
    sig = {
        "mnt753_init": (i, [i]),
        "mnt753_init_devices": (i, [i]),
        "mnt753_device_count": (i, []),
        "mnt753_set_device": (i, [i]),
        "mnt753_get_device": (i, []),
        "mnt753_enable_peer_access": (i, [i, i, C.POINTER(C.c_int)]),
        "mnt753_copy_peer": (i, [i, vp, i, vp, sz]),
        "mnt753_copy_peer_async": (i, [i, vp, i, vp, sz]),
        "mnt753_last_error": (C.c_char_p, []),
        "mnt753_exchange_points": (i, [C.POINTER(vp), sz, u64p]),
        "mnt753_exchange_last_us": (C.c_double, []),
        "mnt753_affine_words": (sz, [i, i]),
        "mnt753_projective_words": (sz, [i, i]),
        "mnt753_dev_alloc": (i, [C.POINTER(vp), sz]),
        "mnt753_dev_free": (i, [vp]),
        "mnt753_copy_h2d": (i, [vp, vp, sz]),
        "mnt753_copy_d2h": (i, [vp, vp, sz]),
        "mnt753_copy_d2d": (i, [vp, vp, sz]),
        "mnt753_dev_memset": (i, [vp, i, sz]),
        "mnt753_sync": (i, [vp]),
        "mnt753_load_file_to_device": (i, [C.c_char_p, sz, sz, vp]),
        "mnt753_bases_create": (i, [i, i, vp, i, sz, C.POINTER(vp)]),
        "mnt753_bases_free": (i, [vp]),
        "mnt753_bases_size": (sz, [vp]),
        "mnt753_msm": (i, [vp, sz, vp, i, sz, u64p, vp]),
        "mnt753_msm_start": (i, [vp, sz, vp, i, sz, vp]),
        "mnt753_msm_finish": (i, [vp, u64p]),
        "mnt753_msm_set_window_bits": (i, [i]),
        "mnt753_msm_set_window_table": (i, [i]),
        "mnt753_dev_mem_info": (i, [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
        "mnt753_self_test": (i, [i]),
        "mnt753_self_test_curve": (i, [i, i]),
        "mnt753_msm_order_after": (i, [vp, vp]),
        "mnt753_msm_last_timing": (i, [C.POINTER(C.c_float)]),
        "mnt753_msm_last_plan": (i, [C.POINTER(C.c_int)]),
        "mnt753_msm_last_pair_levels": (i, []),
        "mnt753_msm_last_irr_levels": (i, []),
        "mnt753_point_add": (i, [i, i, u64p, u64p, u64p]),
        "mnt753_point_scale": (i, [i, i, u64p, u64p, u64p]),
        "mnt753_point_to_affine": (i, [i, i, u64p, u64p]),
        "mnt753_point_from_affine": (i, [i, i, u64p, u64p]),
        "mnt753_domain_create": (i, [i, sz, C.POINTER(vp)]),
        "mnt753_domain_free": (i, [vp]),
        "mnt753_domain_size": (sz, [vp]),
        "mnt753_fft": (i, [vp, i, vp, vp]),
        "mnt753_divide_by_z_on_coset": (i, [vp, vp, vp]),
        "mnt753_vec_muleq": (i, [i, vp, vp, sz, vp]),
        "mnt753_vec_subeq": (i, [i, vp, vp, sz, vp]),
        "mnt753_vec_scale": (i, [i, vp, vp, vp, sz, vp]),
        "mnt753_compute_h": (i, [vp, vp, vp, vp, vp, vp]),
        "mnt753_compute_h_chain": (i, [vp, vp, vp]),
        "mnt753_compute_h_finish": (i, [vp, vp, vp, vp, vp, vp]),
        "mnt753_domain_device": (i, [vp]),
        "mnt753_synth_scalars": (i, [i, C.c_uint64, sz, u64p]),
        "mnt753_r1cs_create": (i, [i, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
        "mnt753_r1cs_free": (i, [vp]),
        "mnt753_r1cs_domain_size": (sz, [vp]),
        "mnt753_r1cs_num_variables": (sz, [vp]),
        "mnt753_r1cs_num_inputs": (sz, [vp]),
        "mnt753_r1cs_evaluate": (i, [vp, vp, vp, vp, vp, sz, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)   # AttributeError here = the library does not export what the header declares
        fn.restype, fn.argtypes = res, args
    _LIB = L
    return L


_TEST_LIB = None


def test_lib_path():
    return os.environ.get("MNT753_TEST_LIB") or os.path.join(_HERE, "libmnt753_hip_test.so")


def test_lib():
    """libmnt753_hip_test.so (include/mnt753_hip_test.h): synthetic bases with known discrete logarithms and the device-level
    known-answer hooks.  Test infrastructure: tests/, bench.py and __graft_entry__.smoke() load it, the product never does."""
    global _TEST_LIB
    if _TEST_LIB is not None:
        return _TEST_LIB
    lib()   # the product first: the test library resolves set_error / require_device and the HIP state against it
    path = test_lib_path()
    if not os.path.exists(path):
        raise Mnt753Error(f"{path} not found: build the HIP extension first (make, or __graft_entry__.build())")
    L = C.CDLL(path)
    u64p, sz, i = C.POINTER(C.c_uint64), C.c_size_t, C.c_int
    sig = {
        "mnt753_synth_points": (i, [i, i, C.c_uint64, sz, u64p, i]),
        "mnt753_synth_expected_msm": (i, [i, i, C.c_uint64, sz, u64p, u64p]),
        "mnt753_test_field_op": (i, [i, i, u64p, u64p, sz, u64p]),
        "mnt753_test_ext_op": (i, [i, i, i, u64p, u64p, sz, u64p]),
        "mnt753_test_point_op": (i, [i, i, i, i, u64p, u64p, sz, u64p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _TEST_LIB = L
    return L


def _check(rc, what):
    if rc != 0:
        raise Mnt753Error(f"{what} failed (rc={rc}): {lib().mnt753_last_error().decode()}")


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint64))


def init(device=0):
    _check(lib().mnt753_init(int(device)), "mnt753_init")


def self_test(level=1, curve=None):
    """mnt753_self_test / _curve: the known answers embedded in the library (include/mnt753_hip.h); raises Mnt753Error on a mismatch"""
    if curve is None:
        _check(lib().mnt753_self_test(level), "mnt753_self_test")
    else:
        _check(lib().mnt753_self_test_curve(curve, level), "mnt753_self_test")


def exchange_points(blocks):
    """all-gather of one block of u64 words per logical device over RCCL inside the boundary (mnt753_exchange_points);
    returns (list of blocks in rank order, microseconds)."""
    keep = [np.ascontiguousarray(b, dtype=np.uint64) for b in blocks]
    words = keep[0].size
    arr = (C.c_void_p * len(keep))(*[C.c_void_p(k.ctypes.data) for k in keep])
    out = np.zeros(len(keep) * words, dtype=np.uint64)
    _check(lib().mnt753_exchange_points(arr, words, out.ctypes.data_as(C.POINTER(C.c_uint64))), "mnt753_exchange_points")
    return [out[g * words:(g + 1) * words].copy() for g in range(len(keep))], float(lib().mnt753_exchange_last_us())


def affine_words(curve, group):
    return int(lib().mnt753_affine_words(curve, group))


def projective_words(curve, group):
    return int(lib().mnt753_projective_words(curve, group))


class BaseSet:
    """Device-resident vector_G1 / vector_G2 (B::params_A ... of the reference)."""

    def __init__(self, curve, group, affine, on_device=False, n=None):
        self.curve, self.group = curve, group
        self._h = C.c_void_p()
        if on_device:
            ptr, cnt = C.c_void_p(int(affine)), int(n)
        else:
            self._keep = np.ascontiguousarray(affine, dtype=np.uint64)
            cnt = self._keep.size // affine_words(curve, group) if n is None else int(n)
            ptr = C.c_void_p(self._keep.ctypes.data)
        _check(lib().mnt753_bases_create(curve, group, ptr, 1 if on_device else 0, cnt, C.byref(self._h)), "mnt753_bases_create")
        self.n = cnt

    def msm(self, scalars, n=None, base_offset=0, on_device=False, stream=None):
        """sum scalars[i] * bases[base_offset + i]; returns the projective wire words (numpy u64)."""
        out = np.zeros(projective_words(self.curve, self.group), dtype=np.uint64)
        if on_device:
            sptr, cnt = C.c_void_p(int(scalars)), int(n)
        else:
            keep = np.ascontiguousarray(scalars, dtype=np.uint64)
            cnt = keep.size // 12 if n is None else int(n)
            sptr = C.c_void_p(keep.ctypes.data)
        st = C.c_void_p(int(stream)) if stream else C.c_void_p()
        _check(lib().mnt753_msm(self._h, base_offset, sptr, 1 if on_device else 0, cnt,
                                out.ctypes.data_as(C.POINTER(C.c_uint64)), st), "mnt753_msm")
        return out

    def msm_start(self, scalars_dev_ptr, n, base_offset=0, stream=None):
        """Enqueue sum scalars[i] * bases[base_offset + i] (scalars resident on the device); collect with msm_finish()."""
        st = C.c_void_p(int(stream)) if stream else C.c_void_p()
        _check(lib().mnt753_msm_start(self._h, base_offset, C.c_void_p(int(scalars_dev_ptr)), 1, int(n), st), "mnt753_msm_start")

    def order_after(self, first):
        """The point kernels of this set's NEXT msm_start begin when those of `first`'s MSM in flight have ended (mnt753_msm_order_after)."""
        _check(lib().mnt753_msm_order_after(self._h, first._h), "mnt753_msm_order_after")

    def msm_finish(self):
        out = np.zeros(projective_words(self.curve, self.group), dtype=np.uint64)
        _check(lib().mnt753_msm_finish(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64))), "mnt753_msm_finish")
        return out

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().mnt753_bases_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def msm_last_timing():
    t = (C.c_float * 5)()
    _check(lib().mnt753_msm_last_timing(t), "mnt753_msm_last_timing")
    return dict(total_ms=t[0], sort_ms=t[1], accumulate_ms=t[2], reduce_ms=t[3], host_ms=t[4])


def msm_last_plan():
    t = (C.c_int * 4)()
    _check(lib().mnt753_msm_last_plan(t), "mnt753_msm_last_plan")
    return dict(window_bits=t[0], windows=t[1], window_table=bool(t[2]), entries_per_lane=t[3],
                pair_levels=int(lib().mnt753_msm_last_pair_levels()), irr_levels=int(lib().mnt753_msm_last_irr_levels()))


def point_add(curve, group, a, b):
    a, pa = _u64(a); b, pb = _u64(b)
    out = np.zeros(projective_words(curve, group), dtype=np.uint64)
    _check(lib().mnt753_point_add(curve, group, pa, pb, out.ctypes.data_as(C.POINTER(C.c_uint64))), "mnt753_point_add")
    return out


def point_scale(curve, group, scalar, p):
    s, ps = _u64(scalar); p, pp = _u64(p)
    out = np.zeros(projective_words(curve, group), dtype=np.uint64)
    _check(lib().mnt753_point_scale(curve, group, ps, pp, out.ctypes.data_as(C.POINTER(C.c_uint64))), "mnt753_point_scale")
    return out


def point_to_affine(curve, group, p):
    p, pp = _u64(p)
    out = np.zeros(affine_words(curve, group), dtype=np.uint64)
    _check(lib().mnt753_point_to_affine(curve, group, pp, out.ctypes.data_as(C.POINTER(C.c_uint64))), "mnt753_point_to_affine")
    return out


def point_from_affine(curve, group, a):
    a, pa = _u64(a)
    out = np.zeros(projective_words(curve, group), dtype=np.uint64)
    _check(lib().mnt753_point_from_affine(curve, group, pa, out.ctypes.data_as(C.POINTER(C.c_uint64))), "mnt753_point_from_affine")
    return out


class DeviceBuffer:
    """Device memory owned through the C ABI (for callers without torch)."""

    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        self.nbytes = int(nbytes)
        _check(lib().mnt753_dev_alloc(C.byref(self.ptr), self.nbytes), "mnt753_dev_alloc")

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        _check(lib().mnt753_copy_h2d(b.ptr, C.c_void_p(a.ctypes.data), a.nbytes), "mnt753_copy_h2d")
        return b

    @classmethod
    def from_file(cls, path, offset, nbytes):
        """Stream nbytes of a file (from byte offset) into new device memory (mnt753_load_file_to_device)."""
        b = cls(nbytes)
        _check(lib().mnt753_load_file_to_device(str(path).encode(), int(offset), int(nbytes), b.ptr), "mnt753_load_file_to_device")
        return b

    def to_numpy(self, dtype=np.uint64):
        out = np.zeros(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        _check(lib().mnt753_copy_d2h(C.c_void_p(out.ctypes.data), self.ptr, self.nbytes), "mnt753_copy_d2h")
        return out

    def close(self):
        if self.ptr and self.ptr.value:
            lib().mnt753_dev_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Domain:
    """basic_radix2_domain over Fr of the curve (B::get_evaluation_domain)."""

    def __init__(self, curve, m):
        self.curve, self.m = curve, int(m)
        self._h = C.c_void_p()
        _check(lib().mnt753_domain_create(curve, self.m, C.byref(self._h)), "mnt753_domain_create")

    def fft(self, kind, dev_ptr, stream=None):
        st = C.c_void_p(int(stream)) if stream else C.c_void_p()
        _check(lib().mnt753_fft(self._h, kind, C.c_void_p(int(dev_ptr)), st), "mnt753_fft")

    def divide_by_z_on_coset(self, dev_ptr, stream=None):
        st = C.c_void_p(int(stream)) if stream else C.c_void_p()
        _check(lib().mnt753_divide_by_z_on_coset(self._h, C.c_void_p(int(dev_ptr)), st), "mnt753_divide_by_z_on_coset")

    def compute_h(self, ca, cb, cc, h, stream=None):
        st = C.c_void_p(int(stream)) if stream else C.c_void_p()
        _check(lib().mnt753_compute_h(self._h, C.c_void_p(int(ca)), C.c_void_p(int(cb)), C.c_void_p(int(cc)),
                                      C.c_void_p(int(h)), st), "mnt753_compute_h")

    def compute_h_chain(self, vec, stream=None):
        """vec <- cosetFFT(iFFT(vec)): the per-vector half of compute_H (runs on the domain's device)."""
        st = C.c_void_p(int(stream)) if stream else C.c_void_p()
        _check(lib().mnt753_compute_h_chain(self._h, C.c_void_p(int(vec)), st), "mnt753_compute_h_chain")

    def compute_h_finish(self, a, b, c, h, stream=None):
        """a <- icosetFFT((a * b - c) / Z), h <- a | 0: the joining half of compute_H."""
        st = C.c_void_p(int(stream)) if stream else C.c_void_p()
        _check(lib().mnt753_compute_h_finish(self._h, C.c_void_p(int(a)), C.c_void_p(int(b)), C.c_void_p(int(c)), C.c_void_p(int(h)), st),
               "mnt753_compute_h_finish")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().mnt753_domain_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def vec_muleq(curve, a_ptr, b_ptr, n, stream=None):
    st = C.c_void_p(int(stream)) if stream else C.c_void_p()
    _check(lib().mnt753_vec_muleq(curve, C.c_void_p(int(a_ptr)), C.c_void_p(int(b_ptr)), n, st), "mnt753_vec_muleq")


def copy_d2d(dst_ptr, src_ptr, nbytes):
    """Asynchronous device-to-device copy on the default stream (mnt753_copy_d2d)."""
    _check(lib().mnt753_copy_d2d(C.c_void_p(int(dst_ptr)), C.c_void_p(int(src_ptr)), int(nbytes)), "mnt753_copy_d2d")


def vec_scale(curve, dst_ptr, src_ptr, scalar, n, stream=None):
    """dst[i] = src[i] * scalar (scalar: 12 uint64 on the host, wire format)."""
    st = C.c_void_p(int(stream)) if stream else C.c_void_p()
    k = np.ascontiguousarray(scalar, dtype=np.uint64)
    assert k.size == 12
    _check(lib().mnt753_vec_scale(curve, C.c_void_p(int(dst_ptr)), C.c_void_p(int(src_ptr)), k.ctypes.data_as(C.c_void_p), n, st), "mnt753_vec_scale")


def vec_subeq(curve, a_ptr, b_ptr, n, stream=None):
    st = C.c_void_p(int(stream)) if stream else C.c_void_p()
    _check(lib().mnt753_vec_subeq(curve, C.c_void_p(int(a_ptr)), C.c_void_p(int(b_ptr)), n, st), "mnt753_vec_subeq")


def synth_points(curve, group, seed, n, threads=None):
    out = np.zeros((n, affine_words(curve, group)), dtype=np.uint64)
    threads = threads or min(64, os.cpu_count() or 1)
    _check(test_lib().mnt753_synth_points(curve, group, seed, n, out.ctypes.data_as(C.POINTER(C.c_uint64)), threads), "mnt753_synth_points")
    return out


def synth_scalars(curve, seed, n):
    out = np.zeros((n, 12), dtype=np.uint64)
    _check(lib().mnt753_synth_scalars(curve, seed, n, out.ctypes.data_as(C.POINTER(C.c_uint64))), "mnt753_synth_scalars")
    return out


def synth_expected_msm(curve, group, seed, scalars):
    s, ps = _u64(scalars)
    out = np.zeros(projective_words(curve, group), dtype=np.uint64)
    _check(test_lib().mnt753_synth_expected_msm(curve, group, seed, s.size // 12, ps, out.ctypes.data_as(C.POINTER(C.c_uint64))), "mnt753_synth_expected_msm")
    return out


def read_r1cs_file(path):
    """r1cs.bin of oracle/ref_groth16.cpp: u64 num_inputs, m, nc; per matrix a, b, c: u64 row_ptr[nc + 1], u32 col[nnz], Fr coeff[nnz]."""
    raw = np.fromfile(path, dtype=np.uint8)
    num_inputs, m, nc = (int(v) for v in raw[:24].view(np.uint64))
    pos, mats = 24, []
    for _ in range(3):
        rp = raw[pos:pos + 8 * (nc + 1)].view(np.uint64).copy(); pos += 8 * (nc + 1)
        nnz = int(rp[nc])
        col = raw[pos:pos + 4 * nnz].view(np.uint32).copy(); pos += 4 * nnz
        cf = raw[pos:pos + 96 * nnz].view(np.uint64).copy().reshape(nnz, 12); pos += 96 * nnz
        mats.append((rp, col, cf))
    assert pos == raw.size
    return num_inputs, m, nc, mats


class R1cs:
    """Device-resident constraint system (mnt753_r1cs_*): the witness-map front end."""

    def __init__(self, curve, num_inputs, m, nc, mats):
        self.curve, self.num_inputs, self.m, self.nc = curve, num_inputs, m, nc
        self._h = C.c_void_p()
        self._keep = [(np.ascontiguousarray(rp, dtype=np.uint64), np.ascontiguousarray(col, dtype=np.uint32), np.ascontiguousarray(cf, dtype=np.uint64)) for rp, col, cf in mats]
        arr = lambda k: (C.c_void_p * 3)(*[C.c_void_p(t[k].ctypes.data) for t in self._keep])
        _check(lib().mnt753_r1cs_create(curve, num_inputs, m, nc, arr(0), arr(1), arr(2), C.byref(self._h)), "mnt753_r1cs_create")

    @classmethod
    def from_file(cls, curve, path):
        return cls(curve, *read_r1cs_file(path))

    def domain_size(self):
        return int(lib().mnt753_r1cs_domain_size(self._h))

    def evaluate(self, dev_w, dev_ca, dev_cb, dev_cc, out_len, stream=None):
        st = C.c_void_p(int(stream)) if stream else C.c_void_p()
        _check(lib().mnt753_r1cs_evaluate(self._h, C.c_void_p(int(dev_w)), C.c_void_p(int(dev_ca)), C.c_void_p(int(dev_cb)), C.c_void_p(int(dev_cc)),
                                          int(out_len), st), "mnt753_r1cs_evaluate")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().mnt753_r1cs_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def test_field_op(mod, op, a, b=None):
    """Test hook: the device field layer element-wise on wire-form elements (see include/mnt753_hip.h)."""
    a, pa = _u64(a)
    b, pb = _u64(a if b is None else b)
    out = np.zeros_like(a)
    _check(test_lib().mnt753_test_field_op(mod, op, pa, pb, a.size // 12, out.ctypes.data_as(C.POINTER(C.c_uint64))), "mnt753_test_field_op")
    return out


def test_ext_op(curve, split, op, a, b=None):
    """Test hook: Fq2 (MNT4753) / Fq3 (MNT6753) on the device, element-wise; split = the lane-split form of the G2 kernels."""
    a, pa = _u64(a)
    b, pb = _u64(a if b is None else b)
    out = np.zeros_like(a)
    words = 12 * (2 if curve == 0 else 3)
    _check(test_lib().mnt753_test_ext_op(curve, int(split), op, pa, pb, a.size // words, out.ctypes.data_as(C.POINTER(C.c_uint64))), "mnt753_test_ext_op")
    return out


def test_point_op(curve, group, split, op, p, q=None):
    """Test hook: one form of the device group law on projective wire points (see include/mnt753_hip.h)."""
    p, pp = _u64(p)
    q, pq = _u64(p if q is None else q)
    out = np.zeros_like(p)
    _check(test_lib().mnt753_test_point_op(curve, group, int(split), op, pp, pq, p.size // projective_words(curve, group),
                                      out.ctypes.data_as(C.POINTER(C.c_uint64))), "mnt753_test_point_op")
    return out


test_ext_op.__test__ = False
test_point_op.__test__ = False


def mont_one(curve):
    """Fr element 1 in wire (Montgomery) form: R mod r."""
    r = [0x0001c4c62d92c41110229022eee2cdadb7f997505b8fafed5eb7e8f96c97d87307fdb925e8a0ed8d99d124d9a15af79db26c5c28c859a99b3eebca9429212636b9dff97634993aa4d6c381bc3f0057974ea099170fa13a4fd90776e240000001,
         0x0001c4c62d92c41110229022eee2cdadb7f997505b8fafed5eb7e8f96c97d87307fdb925e8a0ed8d99d124d9a15af79db117e776f218059db80f0da5cb537e38685acce9767254a4638810719ac425f0e39d54522cdd119f5e9063de245e8001][curve]
    v = (1 << 768) % r
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(12)], dtype=np.uint64)
