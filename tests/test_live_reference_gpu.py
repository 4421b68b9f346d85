"""GPU: a LIVE differential test against the reference itself, BASELINE configs[0] literally -- `generate_parameters fast`
(libsnark/generate_parameters.cpp:23-133: a fresh random keypair and the R1CS-chain witness, MNT4753 d + 1 = 2^14 and MNT6753
d + 1 = 2^10, different bytes on every run), then the reference's `./main <curve> compute` (libsnark/main.cpp, bos_coster) and
`main_hip` on the same files; the proofs must be the same bytes (README.md:47-58 of the reference: sha256 of the outputs).

The binaries are the reference's own sources compiled where they lie by oracle/build_ref.sh (oracle/_ref/, git-ignored; they
travel to the GPU box like the built library) -- test infrastructure: nothing of the product links or runs them.

MNT753_REAL_PARAMS=1 (opt-in: minutes of host time) does the same with the generator's FULL sizes, 2^20 / 2^15 -- the "full-size
generated parameters" of the north star; the result of the round's run is recorded in BASELINE.md."""
import filecmp
import os
import shutil
import subprocess
import time

import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
REF = os.path.join(O.ROOT, "oracle", "_ref")
EXE = os.path.join(O.ROOT, "snark-challenge-prover-reference_amd", "main_hip")
PIECEWISE_HIP = os.path.join(REF, "piecewise_hip")


def need_ref():
    O.need_ref("generate_parameters", "main", "piecewise_hip")     # missing = failure on a GPU box (tests/oracle_lib.py)


def generate(work, fast):
    t0 = time.time()
    r = subprocess.run([os.path.join(REF, "generate_parameters")] + (["fast"] if fast else []), cwd=work, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    for f in ("MNT4753-parameters", "MNT4753-input", "MNT6753-parameters", "MNT6753-input"):
        assert os.path.getsize(os.path.join(work, f)) > 0
    return time.time() - t0


def prove_ref(work, curve):
    out = os.path.join(work, f"{curve}-output-ref")
    t0 = time.time()
    r = subprocess.run([os.path.join(REF, "main"), curve, "compute", os.path.join(work, f"{curve}-parameters"), os.path.join(work, f"{curve}-input"), out],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out, time.time() - t0


def prove_hip(work, curve, tag, flags=(), env=None, exe=EXE):
    out = os.path.join(work, f"{curve}-output-{tag}")
    r = subprocess.run([exe, curve, "compute", os.path.join(work, f"{curve}-parameters"), os.path.join(work, f"{curve}-input"), out] + list(flags),
                       capture_output=True, text=True, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-2000:]
    return out, r.stdout


CONFIGS = [("default", [], {}), ("ref_order", ["--ref-order", "--unfused-h"], {}), ("unfused_c", ["--unfused-c"], {}),
           ("gpus3", ["--gpus", "3"], {"MNT753_SHARE_DEVICE": "1"}), ("repeat", ["--repeat", "2"], {})]


def test_fresh_fast_parameter_sets_same_bytes_as_the_reference(gpu, tmp_path):
    need_ref()
    work = str(tmp_path)
    generate(work, fast=True)
    for curve in ("MNT4753", "MNT6753"):
        ref_out, _ = prove_ref(work, curve)
        sizes = {"MNT4753": 768, "MNT6753": 960}
        assert os.path.getsize(ref_out) == sizes[curve]
        for tag, flags, env in CONFIGS:
            out, _ = prove_hip(work, curve, tag, flags, env)
            assert filecmp.cmp(out, ref_out, shallow=False), (curve, tag)
        # the reference's own driver text over the MI355X wrapper (tools/dropin_check.sh)
        out = os.path.join(work, f"{curve}-output-dropin")
        r = subprocess.run([PIECEWISE_HIP, curve, "compute", os.path.join(work, f"{curve}-parameters"), os.path.join(work, f"{curve}-input"), out],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        assert filecmp.cmp(out, ref_out, shallow=False), (curve, "dropin")


@pytest.mark.skipif(os.environ.get("MNT753_REAL_PARAMS") != "1", reason="opt-in (MNT753_REAL_PARAMS=1): the generator's full sizes take minutes of host time")
def test_full_size_generated_parameters_same_bytes_as_the_reference(gpu, tmp_path):
    """2^20 (MNT4753) and 2^15 (MNT6753) from the reference's generator -- not the synthetic files -- through the reference's ./main
    and main_hip.  Prints the timings; the round's run is quoted in BASELINE.md."""
    need_ref()
    work = os.environ.get("MNT753_REAL_PARAMS_DIR") or str(tmp_path)
    os.makedirs(work, exist_ok=True)
    gen_s = generate(work, fast=False)
    report = {"generate_parameters_s": round(gen_s, 1), "host_threads": os.cpu_count()}
    for curve in ("MNT6753", "MNT4753"):
        ref_out, ref_s = prove_ref(work, curve)
        out, stdout = prove_hip(work, curve, "default", ["--repeat", "2"])
        assert filecmp.cmp(out, ref_out, shallow=False), curve
        out2, _ = prove_hip(work, curve, "ref_order", ["--ref-order", "--unfused-h"])
        assert filecmp.cmp(out2, ref_out, shallow=False), curve
        times = [l for l in stdout.splitlines() if "Total time from input to output" in l or l.startswith("load params")]
        report[curve] = {"reference_main_wall_s": round(ref_s, 1), "main_hip": times}
    print("REAL_PARAMS_REPORT", report)
    rep = os.environ.get("MNT753_REAL_PARAMS_REPORT")
    if rep:
        import json
        with open(rep, "w") as f:
            json.dump(report, f, indent=1)
    if not os.environ.get("MNT753_REAL_PARAMS_DIR"):
        shutil.rmtree(work, ignore_errors=True)
