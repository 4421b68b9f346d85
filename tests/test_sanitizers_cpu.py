"""CPU: the host side of the product under AddressSanitizer + UBSan and under ThreadSanitizer (the reference
ships sanitizer builds of its host code; GPU sanitizers do not exist on the pool).  `make asan` / `make tsan` build host/main.cpp and
host/prover_hip_functions.cpp over tools/stub_abi/stub_mnt753.cpp -- a TEST STUB of the C ABI with host memory and no arithmetic --
so what runs here is exactly the host logic: the input loader thread, the per-device slice loaders, the readiness latches, the
sharded start / fold, batch mode, and every error path of the CLI.  Results are not compared with anything (an MSM of the stub
returns the identity); a sanitizer report or a non-zero exit code fails the test."""
import os
import subprocess

import pytest

import golden_io as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAME = {0: "MNT4753", 1: "MNT6753"}


@pytest.fixture(scope="module")
def san():
    r = subprocess.run(["make", "asan", "tsan"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return {k: os.path.join(ROOT, "build", "san", f"main_hip_{k}") for k in ("asan", "tsan")}


def run(exe, args, expect_rc=0, leaks=1):
    env = dict(os.environ, ASAN_OPTIONS=f"detect_leaks={leaks}:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe] + args, capture_output=True, text=True, env=env, timeout=600)
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == expect_rc, (r.returncode, r.stderr[-2000:])
    return r


@pytest.mark.parametrize("kind", ["asan", "tsan"])
@pytest.mark.parametrize("curve", [0, 1])
def test_host_paths_clean_under_sanitizers(san, kind, curve, tmp_path):
    params, inp, _ = G.e2e_paths(curve)
    out = str(tmp_path / "o")
    for flags in ([], ["--repeat", "3"], ["--gpus", "2", "--ref-order"], ["--gpus", "3", "--unfused-h", "--repeat", "2"], ["--gpus", "8"],
                  [inp, str(tmp_path / "o2"), "--gpus", "4"], ["--unfused-c", "--repeat", "2"], ["--gpus", "3", "--unfused-c"], ["--c-last", "--gpus", "2"], ["--ref-order", "--touch-all", "--repeat", "2"], ["--ref-order", "--touch-all", "--gpus", "2"]):
        r = run(san[kind], [NAME[curve], "compute", params, inp, out] + flags)
        assert "Total time from input to output" in r.stdout


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_compute_h_spread_over_devices_clean_under_sanitizers(san, kind, tmp_path):
    """Round 4: with several devices ca / cb / cc are streamed by the loader threads of devices 0 / 1 / 2 (0 / 1 / 0 with two), each
    device transforms its own vector, cb and cc travel to device 0 for the pointwise step; every device assembles the scalars of its
    part of H | L | B1 itself.  The loader threads, their latches (device 0 releases its range of w, then ca, then the rest of w) and
    the staged peer copies under the sanitizers, for the fused call, the reference's B:: call sequence and uneven device counts."""
    for curve in (0, 1):
        params, inp, _ = G.e2e_paths(curve)
        out = str(tmp_path / "o")
        for n_dev, flags in ((2, []), (3, []), (5, ["--repeat", "2"]), (8, ["--ref-order"]), (2, ["--unfused-h", "--unfused-c"]), (3, ["--ref-order", "--unfused-h"]),
                             (4, ["--unfused-c", "--h-last"]), (3, ["--ref-order", "--touch-all"])):
            env = dict(os.environ, MNT753_TRACE="1", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", TSAN_OPTIONS="halt_on_error=1")
            r = subprocess.run([san[kind], NAME[curve], "compute", params, inp, out, "--gpus", str(n_dev)] + flags, capture_output=True, text=True, env=env, timeout=600)
            assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
            if "--unfused-h" not in flags:
                assert f"compute_H over devices 0 / 1 / {2 if n_dev > 2 else 0}" in r.stderr, r.stderr[-800:]
            if "--unfused-c" not in flags and "--touch-all" not in flags:
                assert f"{n_dev} devices" in r.stderr, r.stderr[-800:]


def test_peer_access_requested_for_every_ordered_pair(san, tmp_path):
    """Round 5: the transformed cb / cc travel 1 -> 0 and 2 -> 0, the slices of coefficients_for_H 0 -> g and operands of B:: vector
    calls between any two devices; the wrapper asks the C ABI for peer access for EVERY ordered pair of its devices (the stub records
    each request) and says per pair what it got (cuda_prover_piecewise.cu:24-34 keeps everything on one device: nothing to ask there)."""
    params, inp, _ = G.e2e_paths(1)
    out = str(tmp_path / "o")
    for n_dev in range(2, 9):
        env = dict(os.environ, MNT753_TRACE="1", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
        r = subprocess.run([san["asan"], "MNT6753", "compute", params, inp, out, "--gpus", str(n_dev)], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0 and "Sanitizer" not in r.stderr, r.stderr[-3000:]
        for a in range(n_dev):
            for b in range(n_dev):
                if a != b:
                    assert f"stub: peer access requested {a} -> {b}\n" in r.stderr, (n_dev, a, b)
                    assert f"mnt753: device {a} reads device {b}: same GPU" in r.stderr, (n_dev, a, b)
        assert r.stderr.count("stub: peer access requested") == n_dev * (n_dev - 1)
    r = subprocess.run([san["asan"], "MNT6753", "compute", params, inp, out], capture_output=True, text=True, env=dict(os.environ, MNT753_TRACE="1"), timeout=600)
    assert r.returncode == 0 and "peer access" not in r.stderr


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_fold_over_the_exchange_clean_under_sanitizers(san, kind, tmp_path):
    """--fold rccl: the partial points of every sharded multiexp go through mnt753_exchange_points (the stub copies them; with
    MNT753_STUB_NO_RCCL=1 it refuses like a box without librccl and the wrapper falls back to the host fold)."""
    params, inp, _ = G.e2e_paths(0)
    out = str(tmp_path / "o")
    for n_dev, extra_env, expect in ((1, {}, "over RCCL"), (3, {}, "over RCCL"), (8, {}, "over RCCL"), (2, {"MNT753_STUB_NO_RCCL": "1"}, "folded on the host")):
        env = dict(os.environ, MNT753_TRACE="1", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", TSAN_OPTIONS="halt_on_error=1", **extra_env)
        r = subprocess.run([san[kind], "MNT4753", "compute", params, inp, out, "--gpus", str(n_dev), "--fold", "rccl", "--repeat", "2"], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
        assert expect in r.stderr, r.stderr[-800:]
    r = run(san[kind], ["MNT4753", "compute", params, inp, out, "--gpus", "2", "--fold", "host"])
    assert "over RCCL" not in r.stderr


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_resident_job_feed_clean_under_sanitizers(san, kind, tmp_path):
    """main_hip --serve: jobs read from stdin against resident parameters, a failing job in the middle does not end the service."""
    params, inp, _ = G.e2e_paths(1)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", TSAN_OPTIONS="halt_on_error=1")
    feed = f"{inp} {tmp_path / 'a'}\n/nonexistent {tmp_path / 'b'}\n{inp} {tmp_path / 'c'}\n"
    r = subprocess.run([san[kind], "MNT6753", "compute", params, inp, str(tmp_path / "o"), "--serve", "--quiet", "--gpus", "2"], input=feed, capture_output=True,
                       text=True, env=env, timeout=600)
    assert r.returncode == 0 and "Sanitizer" not in r.stderr, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert [l.split()[0] for l in lines] == ["proved", "failed", "proved"]


@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_r1cs_front_end_and_completion_clean_under_sanitizers(san, kind, tmp_path):
    for curve in (0, 1):
        d = os.path.join(G.GOLDEN, f"g16_mnt{4 if curve == 0 else 6}")
        out = str(tmp_path / "o")
        run(san[kind], [NAME[curve], "compute-r1cs", os.path.join(d, "params.bin"), os.path.join(d, "r1cs.bin"), os.path.join(d, "witness.bin"), out, "--gpus", "2"])
        run(san[kind], [NAME[curve], "complete", os.path.join(d, "keys.bin"), os.path.join(d, "input.bin"), os.path.join(d, "challenge.bin"), str(tmp_path / "full")])


def test_one_shot_policy_reaches_the_c_abi(san, tmp_path):
    """host/main.cpp: one job on one device without --repeat / --serve is a one-proof process like the reference's CLI
    (libsnark/main.cpp:274-293) -- B::one_shot(true), i.e. mnt753_msm_set_window_table(0) around read_params and no warm-up MSM; several
    jobs, --repeat, --serve, --gpus or --tables keep the tables; --one-shot forces them off.  The stub logs the mode it is handed."""
    params, inp, _ = G.e2e_paths(1)
    out = str(tmp_path / "o")
    def modes(flags, extra_env=None):
        env = dict(os.environ, MNT753_STUB_LOG="1", ASAN_OPTIONS="detect_leaks=1", **(extra_env or {}))
        r = subprocess.run([san["asan"], "MNT6753", "compute", params, inp, out] + flags, capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0 and "Sanitizer" not in r.stderr, r.stderr[-2000:]
        first = [l for l in r.stderr.splitlines() if l.startswith("stub: window table mode")]
        return (first[0].rsplit(" ", 1)[1] if first else None), r.stdout
    m, so = modes([])
    assert m == "0" and "one-shot prover" in so
    for flags in (["--repeat", "2"], ["--tables"], ["--gpus", "2"], [inp, str(tmp_path / "o2")]):
        m, so = modes(flags)
        assert m == "1" and "one-shot prover" not in so, flags
    m, so = modes(["--one-shot", "--repeat", "2"])
    assert m == "0"
    m, so = modes([], {"MNT753_ONE_SHOT": "0"})
    assert m == "1"


def test_self_test_is_run_once_and_is_fatal(san, tmp_path):
    """B::init_public_params runs mnt753_self_test(1) once per process (the stub logs the call; the arithmetic behind it is a GPU test,
    tests/test_selftest_gpu.py); a failure ends the prover before it reads a parameter; MNT753_SELFTEST=0 skips it."""
    params, inp, _ = G.e2e_paths(0)
    out = str(tmp_path / "o")
    def go(extra):
        env = dict(os.environ, MNT753_STUB_LOG="1", ASAN_OPTIONS="detect_leaks=0", **extra)
        return subprocess.run([san["asan"], "MNT4753", "compute", params, inp, out, "--repeat", "2"], capture_output=True, text=True, env=env, timeout=600)
    r = go({})
    assert r.returncode == 0 and r.stderr.count("stub: self-test level 1 curve 0") == 1, r.stderr[-800:]
    r = go({"MNT753_SELFTEST": "0"})
    assert r.returncode == 0 and "stub: self-test" not in r.stderr
    r = go({"MNT753_STUB_SELFTEST_FAILS": "1"})
    assert r.returncode == 1 and "mnt753_self_test" in r.stderr and "Sanitizer" not in r.stderr, r.stderr[-800:]
    assert "load params" not in r.stdout


def test_error_paths_clean_under_asan(san, tmp_path):
    """Every way the CLI can fail: no invalid access on the way out.  Leak checking is off here only: the wrapper keeps the
    reference's raw-pointer interface (B::read_params returns a `new`-ed object the driver deletes at the end), so an exception that
    ends the process leaves the objects built so far to the operating system -- as the reference's own driver does."""
    import functools
    run_err = functools.partial(run, leaks=0)
    exe = san["asan"]
    params, inp, _ = G.e2e_paths(0)
    out = str(tmp_path / "o")
    run_err(exe, ["MNT4753", "compute", "/nonexistent", inp, out], expect_rc=1)
    run_err(exe, ["MNT4753", "compute", params, "/nonexistent", out], expect_rc=1)
    run_err(exe, ["BN128", "compute", params, inp, out], expect_rc=2)
    run_err(exe, ["MNT4753"], expect_rc=2)
    run_err(exe, ["MNT4753", "compute", params, inp, out, "--gpus", "99"], expect_rc=1)
    # an option this prover does not have (--point-cus left in round 5), an option that lost its argument, an input without its output:
    # refused with exit code 2, not parsed as a job pair or silently dropped (round-5 advice)
    r = run_err(exe, ["MNT4753", "compute", params, inp, out, "--point-cus", "240"], expect_rc=2)
    assert "unknown option --point-cus" in r.stderr
    run_err(exe, ["MNT4753", "compute", params, inp, out, "--repeat"], expect_rc=2)
    r = run_err(exe, ["MNT4753", "compute", params, inp, out, inp], expect_rc=2)
    assert "without an output path" in r.stderr
    raw = bytearray(open(params, "rb").read())
    (tmp_path / "trunc").write_bytes(bytes(raw[:-8]))
    run_err(exe, ["MNT4753", "compute", str(tmp_path / "trunc"), inp, out], expect_rc=1)
    raw[8:16] = (1 << 40).to_bytes(8, "little")
    (tmp_path / "badhdr").write_bytes(bytes(raw))
    run_err(exe, ["MNT4753", "compute", str(tmp_path / "badhdr"), inp, out], expect_rc=1)
    short_in = open(inp, "rb").read()[:-96]
    (tmp_path / "short_in").write_bytes(short_in)
    run_err(exe, ["MNT4753", "compute", params, str(tmp_path / "short_in"), out, "--gpus", "2"], expect_rc=1)
    # a constraint system whose variable count does not match the parameters (the out-of-bounds read of the round-2 advice)
    d = os.path.join(G.GOLDEN, "g16_mnt4")
    cs = bytearray(open(os.path.join(d, "r1cs.bin"), "rb").read())
    cs[8:16] = (int.from_bytes(cs[8:16], "little") + 5).to_bytes(8, "little")
    (tmp_path / "cs_bad").write_bytes(bytes(cs))
    r = run_err(exe, ["MNT4753", "compute-r1cs", os.path.join(d, "params.bin"), str(tmp_path / "cs_bad"), os.path.join(d, "witness.bin"), out], expect_rc=1)
    assert "variables" in r.stderr
