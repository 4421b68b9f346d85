"""MI355X-native Groth16 prover hot path (MSM + FFT over MNT4753 / MNT6753).

The product is the HIP library `libmnt753_hip.so` (csrc/, C ABI in include/mnt753_hip.h) and the C++
host mirror of the reference's `B::` wrapper (include/prover_hip_functions.hpp).  This Python package
is only the ctypes face of that C ABI, used by tests/, bench.py and __graft_entry__.py; the synthetic base points and the
device-level test hooks it also exposes come from the TEST library libmnt753_hip_test.so (include/mnt753_hip_test.h).
"""
from .api import (  # noqa: F401
    CURVE_MNT4753, CURVE_MNT6753, G1, G2, FFT, IFFT, COSET_FFT, ICOSET_FFT,
    Mnt753Error, lib, lib_path, test_lib, init, BaseSet, Domain, affine_words, projective_words,
    point_add, point_scale, point_to_affine, point_from_affine, vec_muleq, vec_subeq, vec_scale, copy_d2d,
    synth_points, synth_scalars, synth_expected_msm, msm_last_timing, msm_last_plan, DeviceBuffer, R1cs, read_r1cs_file, self_test,
)
from . import parallel  # noqa: F401,E402
from . import api  # noqa: F401,E402
