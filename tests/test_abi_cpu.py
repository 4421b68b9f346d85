"""CPU: the C-ABI library loads, exports everything include/mnt753_hip.h declares, refuses to compute without a
GPU (no silent fallback), and its host-side group helpers / synthetic generators agree with the goldens and oracle."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import golden_io as G
import oracle_lib as O

ROOT = O.ROOT


def header_symbols(name="mnt753_hip.h"):
    text = open(os.path.join(ROOT, "include", name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mnt753_[a-z0-9_]+)\s*\(", text)))


def exported(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {l.split()[-1] for l in out.splitlines() if l.split()[-1].startswith("mnt753_")}


def test_library_exports_every_declared_symbol(pkg):
    L = ctypes.CDLL(pkg.lib_path())
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/mnt753_hip.h but not exported"
    pkg.lib()   # the ctypes signature table resolves as well


def test_product_and_test_library_are_separate(pkg):
    """Round 5: the product library exports exactly what include/mnt753_hip.h declares -- no test hook, no synthetic base
    generator; those are libmnt753_hip_test.so (include/mnt753_hip_test.h), which needs the product, never the other way round."""
    prod, test = exported(pkg.lib_path()), exported(pkg.api.test_lib_path())
    assert prod == set(header_symbols()), sorted(prod ^ set(header_symbols()))
    assert test == set(header_symbols("mnt753_hip_test.h")), sorted(test)
    assert not any(s.startswith("mnt753_test_") or s in ("mnt753_synth_points", "mnt753_synth_expected_msm") for s in prod)
    needed = lambda p: subprocess.run(["readelf", "-d", p], capture_output=True, text=True, check=True).stdout
    assert "libmnt753_hip.so" in needed(pkg.api.test_lib_path())
    assert "libmnt753_hip_test" not in needed(pkg.lib_path())
    assert "libmnt753_hip_test" not in needed(os.path.join(ROOT, "snark-challenge-prover-reference_amd", "main_hip"))
    assert "oracle" not in needed(pkg.lib_path()) and "oracle" not in needed(pkg.api.test_lib_path())
    pkg.api.test_lib()


def test_sizes(pkg):
    assert [pkg.affine_words(c, g) for c in (0, 1) for g in (1, 2)] == [24, 48, 24, 72]
    assert [pkg.projective_words(c, g) for c in (0, 1) for g in (1, 2)] == [36, 72, 36, 108]


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(has_gpu(), reason="checks the no-device behaviour")
def test_no_cpu_fallback(pkg):
    L = pkg.lib()
    assert L.mnt753_init(0) == -2                      # MNT753_ENODEV
    assert b"no HIP device" in L.mnt753_last_error()
    h = ctypes.c_void_p()
    aff = np.zeros(24, dtype=np.uint64)
    assert L.mnt753_bases_create(0, 1, ctypes.c_void_p(aff.ctypes.data), 0, 1, ctypes.byref(h)) == -2
    assert L.mnt753_domain_create(0, 8, ctypes.byref(h)) == -2
    assert L.mnt753_vec_muleq(0, ctypes.c_void_p(8), ctypes.c_void_p(8), 1, None) == -2
    assert L.mnt753_load_file_to_device(b"/dev/null", 0, 0, None) == -2
    assert L.mnt753_init_devices(2) == -2 and L.mnt753_device_count() == 0 and L.mnt753_set_device(0) == -1
    with pytest.raises(pkg.Mnt753Error):
        pkg.init(0)


@pytest.mark.skipif(has_gpu(), reason="checks the no-device behaviour")
def test_cli_fails_loudly_without_gpu(tmp_path):
    exe = os.path.join(ROOT, "snark-challenge-prover-reference_amd", "main_hip")
    params, inp, _ = G.e2e_paths(0)
    r = subprocess.run([exe, "MNT4753", "compute", params, inp, str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr
    assert not (tmp_path / "o.bin").exists()


def test_bad_arguments(pkg):
    L = pkg.lib()
    out = np.zeros(36, dtype=np.uint64)
    p = out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
    assert L.mnt753_point_add(7, 1, p, p, p) == -1
    assert L.mnt753_point_add(0, 3, p, p, p) == -1
    assert pkg.affine_words(5, 1) == 0


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("group", [1, 2])
def test_host_group_ops_vs_reference_goldens(pkg, curve, group):
    """mnt753_point_{from_affine,add,scale,to_affine} (host code of the product) on the reference's vectors."""
    for r in G.group(curve, group):
        P = pkg.point_from_affine(curve, group, r["P"])
        Q = pkg.point_from_affine(curve, group, r["Q"])
        assert np.array_equal(pkg.point_to_affine(curve, group, pkg.point_add(curve, group, P, Q)), r["sum"])
        assert np.array_equal(pkg.point_to_affine(curve, group, pkg.point_add(curve, group, P, P)), r["dbl"])
        assert np.array_equal(pkg.point_to_affine(curve, group, pkg.point_scale(curve, group, r["s"], P)), r["mul"])
        assert np.array_equal(pkg.point_to_affine(curve, group, P), r["P"])


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("group", [1, 2])
def test_synthetic_inputs(pkg, curve, group):
    """deterministic, thread-count independent, and consistent with their discrete logs (checked by the oracle)."""
    n = 1500 if group == 1 else 1100    # crosses the 1024-point chunk boundary
    a = pkg.synth_points(curve, group, 99, n, threads=1)
    b = pkg.synth_points(curve, group, 99, n, threads=5)
    assert np.array_equal(a, b)
    assert len({bytes(r) for r in a}) == n
    sc = pkg.synth_scalars(curve, 7, n)
    assert np.array_equal(sc, pkg.synth_scalars(curve, 7, n))
    k = 40
    exp = pkg.point_to_affine(curve, group, pkg.synth_expected_msm(curve, group, 99, sc[:k]))
    assert np.array_equal(O.msm(curve, group, a[:k], sc[:k]), exp)
    # across the chunk boundary
    sel = slice(1010, 1040)
    z = np.zeros_like(sc[:1040]); z[sel] = sc[sel]
    exp = pkg.point_to_affine(curve, group, pkg.synth_expected_msm(curve, group, 99, z))
    assert np.array_equal(O.msm(curve, group, a[sel], sc[sel]), exp)


def test_device_field_arithmetic_compiled_for_the_host(tmp_path):
    """fp753.hip.h is __host__ __device__: the fused multipliers of the lane-split extension fields (fp_mul2, fp_mul3) and
    the dedicated squaring (fp_sqr) must agree with compositions of the plain Montgomery product fp_mul -- which the
    oracle-pinned GPU parity tests cover -- and keep their results in [0, 2p).  2000 random cases per modulus.  Also there: the lazy
    arithmetic of the pairing levels, and (round 4) the carry-free butterflies of the NTT -- two stages on raw signed limbs and one
    normalisation -- against the eager formulas on inputs at the edges of their stated ranges, with the limb bounds asserted."""
    import subprocess
    exe = tmp_path / "host_fp_check"
    src = os.path.join(ROOT, "tools", "host_fp_check.cpp")
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", str(exe), src], check=True, timeout=600)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


def test_window_table_doubling_compiled_for_the_host(tmp_path):
    """Round 5: the doubling chain of the window table runs in modified Jacobian coordinates (jac_dbl, curve753.hip.h), not through
    the reference's projective dbl() -- only affine rows leave the kernel.  On the CPU: 2P against libff's own vectors
    (mnt4753_g1.cpp:315-346, mnt4753_g2.cpp:331-362, mnt6753_g2.cpp:337-368 -> tests/golden/group_*.bin), chains of 45 doublings against
    the host field, and the limb ranges the lazy base-field form states, asserted after every doubling."""
    exe = tmp_path / "host_jac_check"
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tools", "host_jac_check.cpp")], check=True, timeout=600)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout + out.stderr


def test_host_field_inversion(tmp_path):
    """The host tails of the product (affine normalisation of the three results of a proof, the fold of partial points) invert through
    host_field.hpp's binary extended Euclid (round 5; Fermat before).  Against libff's Fp_model::invert / Fp2_model / Fp3_model::inverse
    (fp.tcc:641-685) through the minted vectors field_{A,B}.bin and extfield_mnt{4,6}.bin, and against Fermat on 4000 values per
    modulus including 0, 1, p - 1, powers of two and single stored words."""
    exe = tmp_path / "host_inv_check"
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tools", "host_inv_check.cpp")], check=True, timeout=600)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.skipif(not os.path.isfile("/root/reference/cuda_prover_piecewise.cu"), reason="needs the reference tree (build container only)")
def test_reference_driver_compiles_unchanged_against_the_hip_wrapper(tmp_path):
    """The drop-in claim of include/prover_hip_functions.hpp: compute_H<B>, run_prover<B> and main of the reference's
    cuda_prover_piecewise.cu (lines 14-120, untouched) compile and link with B = mnt4753_hip / mnt6753_hip -- only the two
    instantiations and the include line change (tools/dropin_check.sh verifies that nothing else differs)."""
    r = subprocess.run(["sh", os.path.join(ROOT, "tools", "dropin_check.sh")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    exe = os.path.join(ROOT, "oracle", "_ref", "piecewise_hip")
    assert os.access(exe, os.X_OK)
    if not has_gpu():
        # the reference's driver has no error handling of its own: without a device the wrapper's exception ends the process
        params, inp, _ = G.e2e_paths(0)
        p = subprocess.run([exe, "MNT4753", "compute", params, inp, str(tmp_path / "o")], capture_output=True, text=True)
        assert p.returncode != 0 and "no HIP device" in p.stderr
