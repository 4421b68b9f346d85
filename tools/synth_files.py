"""Write a synthetic parameter file + input file in the reference's exact layout
(libsnark/generate_parameters.cpp:60-108) from the library's deterministic generators.  The reference prover
never validates its inputs, so it (and the oracle) accept these files; they stand in for generate_parameters
output on machines that have neither the reference nor its 1.2 GB parameter files."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DEFAULT_SEED = 0x6d696e61   # SURVEY.md section 8d; tests/golden/oracle_hashes.json records the reference's proofs for this seed


def write_files(pkg, curve, log2_d, params_path, input_path, seed=DEFAULT_SEED):
    d = (1 << log2_d) - 1
    m = d + 1
    with open(params_path, "wb") as f:
        np.array([d, m], dtype=np.uint64).tofile(f)
        A = pkg.synth_points(curve, 1, seed + 1, m + 1); A[m] = 0      # identity at the last index, as in real params
        A.tofile(f)
        B1 = pkg.synth_points(curve, 1, seed + 2, m + 1); B1[m] = 0; B1[m - 1] = 0
        B1.tofile(f)
        B2 = pkg.synth_points(curve, 2, seed + 3, m + 1); B2[m] = 0; B2[m - 1] = 0
        B2.tofile(f)
        pkg.synth_points(curve, 1, seed + 4, m - 1).tofile(f)         # L
        pkg.synth_points(curve, 1, seed + 5, d).tofile(f)             # H
    with open(input_path, "wb") as f:
        w = pkg.synth_scalars(curve, seed + 6, m + 1)
        mont_one = pkg.api.mont_one(curve)
        w[0] = mont_one                                                # w[0] = 1 (generate_parameters.cpp:90)
        w.tofile(f)
        for k in range(3):
            pkg.synth_scalars(curve, seed + 7 + k, d + 1).tofile(f)   # ca, cb, cc
        pkg.synth_scalars(curve, seed + 10, 1).tofile(f)              # r
    return d, m


if __name__ == "__main__":
    from __graft_entry__ import load_package
    pkg = load_package()
    curve = {"MNT4753": 0, "MNT6753": 1}[sys.argv[1]]
    d, m = write_files(pkg, curve, int(sys.argv[2]), sys.argv[3], sys.argv[4])
    print(f"wrote d={d} m={m}")
