"""GPU: the known-answer self-test INSIDE the product (mnt753_self_test, csrc/mnt753_selftest.hip).  Its expected words were computed by
the reference's code (tools/gen_selftest_data.py through oracle/_ref: libff's multi_exp_with_mixed_addition, libfqfft's compute_H call
sequence, libff's group classes) and are embedded in libmnt753_hip.so as constants; B::init_public_params runs level 1 once per
process.  Here: it passes at every level on this build, it FAILS -- with MNT753_ESELFTEST and a message naming the check -- when one
expected word is off (MNT753_SELFTEST_CORRUPT flips a bit of check k), and the prover refuses to prove behind a failed self-test."""
import os
import subprocess

import pytest

import golden_io as G
import oracle_lib as O

pytestmark = pytest.mark.gpu
EXE = os.path.join(O.ROOT, "snark-challenge-prover-reference_amd", "main_hip")
PY = ("import sys; sys.path.insert(0, {root!r}); import importlib; api = importlib.import_module('snark-challenge-prover-reference_amd.api'); "
      "api.init(0); api.self_test({level})")


def run_level(level, corrupt=None):
    env = dict(os.environ)
    if corrupt is not None:
        env["MNT753_SELFTEST_CORRUPT"] = str(corrupt)
    return subprocess.run([os.sys.executable, "-c", PY.format(root=O.ROOT, level=level)], capture_output=True, text=True, env=env, timeout=300)


@pytest.mark.parametrize("level", [0, 1, 2])
def test_self_test_passes_on_this_build(gpu, level):
    gpu.api.self_test(level)
    for curve in (0, 1):
        gpu.api.self_test(level, curve=curve)


@pytest.mark.parametrize("level,k", [(1, 0), (1, 7), (1, 8), (1, 9), (1, 11), (1, 12), (1, 13), (1, 16), (1, 17), (2, 10), (2, 11), (2, 25)])
def test_one_wrong_expected_word_is_reported(gpu, level, k):
    """every kind of check at least once: host addition / doubling (0-7), MSM plain and with levels for G1 and G2 (8-11), compute_H
    (12 = MNT4753's at level 1), the other curve (13-17), and at level 2 the window-table forms (10, 11) and the last check (25)"""
    r = run_level(level, corrupt=k)
    assert r.returncode != 0
    assert f"self-test check {k} failed" in r.stderr, r.stderr[-600:]
    assert "mnt753_self_test" in r.stderr


def test_corrupt_index_out_of_range_changes_nothing(gpu):
    assert run_level(1, corrupt=18).returncode == 0      # level 1 has checks 0 .. 17


def test_prover_runs_it_and_refuses_to_prove_behind_a_failure(gpu, tmp_path):
    params, inp, expected = G.e2e_paths(1)
    out = str(tmp_path / "proof.bin")
    r = subprocess.run([EXE, "MNT6753", "compute", params, inp, out], capture_output=True, text=True, env=dict(os.environ, MNT753_TRACE_LOAD="1"))
    assert r.returncode == 0 and "known-answer self-test of this build" in r.stderr, r.stderr[-600:]
    assert open(out, "rb").read() == open(expected, "rb").read()
    os.remove(out)
    # the wrapper tests ITS curve (mnt753_self_test_curve): 4 host checks, then G1 plain / levels (4, 5), G2 (6, 7), compute_H (8)
    r = subprocess.run([EXE, "MNT6753", "compute", params, inp, out], capture_output=True, text=True, env=dict(os.environ, MNT753_SELFTEST_CORRUPT="7"))
    assert r.returncode == 1 and "self-test check 7 failed" in r.stderr and "MNT6753 G2" in r.stderr, r.stderr[-600:]
    assert not os.path.exists(out), "a prover whose self-test failed must not write a proof"
    r = subprocess.run([EXE, "MNT6753", "compute", params, inp, out], capture_output=True, text=True, env=dict(os.environ, MNT753_SELFTEST_CORRUPT="7", MNT753_SELFTEST="0"))
    assert r.returncode == 0      # switched off: nothing runs, nothing can fail
    r = subprocess.run([EXE, "MNT4753", "self-test"], capture_output=True, text=True)
    assert r.returncode == 0 and "all known answers" in r.stdout, r.stderr[-600:]
