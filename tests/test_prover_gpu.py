"""GPU: the whole prover (C++ host mirror of B:: + CLI main_hip) against the reference's own proof files, and
against the oracle on synthetic parameter sets.  The proof file must be byte-identical (README.md:55-58 of the
reference defines parity as sha256 equality of the outputs)."""
import filecmp
import os
import subprocess

import numpy as np
import pytest

import golden_io as G
import oracle_lib as O
import synth_files

pytestmark = pytest.mark.gpu
EXE = os.path.join(O.ROOT, "snark-challenge-prover-reference_amd", "main_hip")
NAME = {0: "MNT4753", 1: "MNT6753"}


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("flags", [[], ["--unfused-h"], ["--ref-order"], ["--unfused-h", "--ref-order"], ["--unfused-c"], ["--c-last"],
                                   ["--unfused-c", "--h-last"], ["--repeat", "2"], ["--ref-order", "--touch-all", "--repeat", "2"],
                                   ["--tables"], ["--one-shot", "--repeat", "2"]])
def test_reference_proof_files(gpu, curve, flags, tmp_path):
    params, inp, expected = G.e2e_paths(curve)
    out = str(tmp_path / "proof.bin")
    r = subprocess.run([EXE, NAME[curve], "compute", params, inp, out] + flags, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Total time from input to output" in r.stdout
    assert filecmp.cmp(out, expected, shallow=False)
    # the one-shot policy (host/main.cpp): one job on one device and no --repeat / --serve = the reference's CLI (libsnark/main.cpp:274-293),
    # no window tables and no warm-up MSM; a resident prover keeps both; --tables / --one-shot force either
    one_shot = ("--one-shot" in flags) or ("--repeat" not in flags and "--tables" not in flags)
    assert ("one-shot prover" in r.stdout) == one_shot, r.stdout


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("env", [{"MNT753_MSM_SORT": "atomic"}, {"MNT753_MSM_SORT": "part"}, {"MNT753_MSM_SORT": "part", "MNT753_MSM_PAIR": "2"},
                                 {"MNT753_EDGE_FLOW_NODES": "0"}, {"MNT753_EDGE_FLOW_NODES": "100000000"}, {"MNT753_MSM_TMIN": "1"},
                                 {"MNT753_MSM_PAIR": "3", "MNT753_MSM_IRR": "2"}, {"MNT753_MSM_PRECOMP": "0"}])
def test_alternative_kernel_paths_write_the_same_proof(gpu, curve, env, tmp_path):
    """The switches the product keeps (round 5 pruned the measured-and-rejected variants): both sort stages, pairing / irregular levels
    forced onto these small sets, every level of the edge merge through the VM form or through lane groups, one entry per accumulate
    lane, no window table.  Every one must reproduce the reference's proof bytes."""
    params, inp, expected = G.e2e_paths(curve)
    out = str(tmp_path / "proof.bin")
    # --tables: a single job would otherwise run as a one-shot prover, without the window tables these switches also have to hold under
    r = subprocess.run([EXE, NAME[curve], "compute", params, inp, out, "--tables"], capture_output=True, text=True, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stderr
    assert filecmp.cmp(out, expected, shallow=False)


@pytest.mark.parametrize("flags,env", [([], {}), (["--ref-order", "--unfused-h"], {}), (["--unfused-c"], {}), (["--gpus", "2"], {"MNT753_SHARE_DEVICE": "1"}),
                                       ([], {"MNT753_MSM_PAIR": "2", "MNT753_MSM_IRR": "1"})])
def test_reference_generator_fast_set_mnt6753(gpu, flags, env, tmp_path):
    """The reference generator's own `fast` size for MNT6753 (generate_parameters.cpp:127-133: d + 1 = 2^10) -- real keys and the real
    R1CS-chain witness of generate_parameters, not the synthetic generator -- against the proof the reference's ./main wrote for it
    (tests/golden/e2e_mnt6_2p10_*.bin, second capture of oracle/mint_golden.cpp)."""
    params, inp, expected = G.e2e_fast_mnt6_paths()
    out = str(tmp_path / "proof.bin")
    r = subprocess.run([EXE, "MNT6753", "compute", params, inp, out] + flags, capture_output=True, text=True, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stderr
    assert filecmp.cmp(out, expected, shallow=False)


@pytest.mark.parametrize("curve,log2_d", [(0, 11), (1, 10)])
def test_synthetic_set_vs_oracle(gpu, curve, log2_d, tmp_path):
    """MNT6753 at 2^10 is the reference's `generate_parameters fast` size."""
    params, inp = str(tmp_path / "params"), str(tmp_path / "input")
    synth_files.write_files(gpu, curve, log2_d, params, inp)
    out_gpu, out_cpu = str(tmp_path / "gpu.bin"), str(tmp_path / "cpu.bin")
    r = subprocess.run([EXE, NAME[curve], "compute", params, inp, out_gpu, "--fused-h"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    O.prove(curve, params, inp, out_cpu)
    assert filecmp.cmp(out_gpu, out_cpu, shallow=False)


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("world", [1, 2])
def test_multi_gpu_driver_reference_proofs(gpu, curve, world, tmp_path):
    """prove_mgpu.py (base vectors sliced over ranks, one all_gather of partial points per proof) reproduces the
    reference's proof bytes.  world = 2 shares the single test GPU over gloo (PROVE_SHARE_GPU=1): same code path as
    one rank per GPU over RCCL except for the backend."""
    import socket
    import sys
    params, inp, expected = G.e2e_paths(curve)
    out = str(tmp_path / "proof.bin")
    script = os.path.join(O.ROOT, "prove_mgpu.py")
    env = dict(os.environ, PROVE_SHARE_GPU="1")
    if world == 1:
        cmd = [sys.executable, script, NAME[curve], "compute", params, inp, out]
    else:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), script, NAME[curve], "compute", params, inp, out]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert filecmp.cmp(out, expected, shallow=False)
    assert '"total_input_to_output_s"' in r.stdout


@pytest.mark.parametrize("curve", [0, 1])
def test_reference_driver_unchanged(gpu, curve, tmp_path):
    """oracle/_ref/piecewise_hip = the reference's own cuda_prover_piecewise.cu driver (lines 14-120, untouched) compiled over
    include/prover_hip_functions.hpp by tools/dropin_check.sh in the build container: same proof bytes as the reference."""
    exe = O.need_ref("piecewise_hip")             # missing = failure on a GPU box (tests/oracle_lib.py)
    params, inp, expected = G.e2e_paths(curve)
    out = str(tmp_path / "proof.bin")
    r = subprocess.run([exe, NAME[curve], "compute", params, inp, out], capture_output=True, text=True, env=dict(os.environ, MNT753_TRACE="1"))
    assert r.returncode == 0, r.stderr
    assert filecmp.cmp(out, expected, shallow=False)
    # ... and its three multiexps over B1 / L / H plus G1_scale and two G1_add ran as ONE MSM over the concatenated set (LazyPoint in
    # host/prover_hip_functions.cpp), without a line of the driver knowing
    assert "one MSM over H | L | B1" in r.stderr
    # ... and, told so through the environment, it proves as the one-shot process it is (no window tables, no warm-up: INTEGRATION.md 1)
    os.remove(out)
    r = subprocess.run([exe, NAME[curve], "compute", params, inp, out], capture_output=True, text=True, env=dict(os.environ, MNT753_ONE_SHOT="1", MNT753_TRACE_LOAD="1"))
    assert r.returncode == 0, r.stderr
    assert "one-shot prover: no window tables" in r.stderr and filecmp.cmp(out, expected, shallow=False)
    r = subprocess.run([exe, NAME[curve], "compute", params, inp, out], capture_output=True, text=True, env=dict(os.environ, MNT753_TRACE="1", MNT753_FUSED_C="0"))
    assert r.returncode == 0, r.stderr
    assert filecmp.cmp(out, expected, shallow=False) and "one MSM" not in r.stderr


@pytest.mark.parametrize("curve", [0, 1])
def test_expression_fusion_from_the_callers_side(gpu, curve):
    """tools/host_tests/lazy_c_test.cpp over the wrapper: C = Ht + Lt + r Bt1 written in the reference's association and in two
    others, and through B::groth16_C, runs as one MSM over H | L | B1 each time (four trace lines); seven other expressions over the
    same unstarted multiexps (factor on the wrong term, partial sums, a value mixed in, a partial vector, a dropped value) are
    evaluated the plain way; everything agrees with the three multiexps computed on their own."""
    exe = os.path.join(O.ROOT, "snark-challenge-prover-reference_amd", "lazy_c_test")
    params, inp, _ = G.e2e_paths(curve)
    r = subprocess.run([exe, NAME[curve], params, inp], capture_output=True, text=True, env=dict(os.environ, MNT753_TRACE="1"), timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
    assert r.stderr.count("one MSM over H | L | B1") == 4, r.stderr


@pytest.mark.parametrize("curve", [0, 1])
def test_resident_parameters_batch_mode(gpu, curve, tmp_path):
    """main_hip proves several (input, output) pairs against parameters that stay resident on the GPU (window tables,
    workspaces, evaluation domain): the reference pays its 0.4-9 s parameter load per process (main.cpp:196-201)."""
    params, inp, expected = G.e2e_paths(curve)
    o1, o2, o3 = (str(tmp_path / f"proof{k}.bin") for k in range(3))
    r = subprocess.run([EXE, NAME[curve], "compute", params, inp, o1, inp, o2, inp, o3], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("Total time from input to output") == 3 and r.stdout.count("load params:") == 1
    for o in (o1, o2, o3):
        assert filecmp.cmp(o, expected, shallow=False)
    r = subprocess.run([EXE, NAME[curve], "compute", params, inp, o1, "--repeat", "2", "--ref-order"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.count("Total time from input to output") == 2
    assert filecmp.cmp(o1, expected, shallow=False)


@pytest.mark.parametrize("curve", [0, 1])
def test_resident_job_feed(gpu, curve, tmp_path):
    """main_hip --serve: the parameters stay resident in HBM and further (input, output) pairs arrive on stdin, one per line; every
    proof is the reference's, a bad job is reported and the service goes on (the 6 s table build of a full-size set is paid once per
    host process, not once per proof)."""
    params, inp, expected = G.e2e_paths(curve)
    outs = [str(tmp_path / f"p{k}.bin") for k in range(3)]
    feed = f"{inp} {outs[1]}\n/nonexistent {tmp_path / 'x'}\n{inp} {outs[2]}\n"
    r = subprocess.run([EXE, NAME[curve], "compute", params, inp, outs[0], "--serve", "--quiet"], input=feed, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert [l.split()[0] for l in lines] == ["proved", "failed", "proved"]
    for o in outs:
        assert filecmp.cmp(o, expected, shallow=False)


@pytest.mark.parametrize("curve", [0, 1])
@pytest.mark.parametrize("n_dev", [2, 3])
def test_sharded_inside_the_boundary(gpu, curve, n_dev, tmp_path):
    """main_hip --gpus N: ONE process, the five parameter vectors cut into N contiguous slices (multiexp.tcc:417-431), one base
    set per device and slice, scalar slices copied device 0 -> device g, partial points folded in rank order -- B::multiexp_G1
    itself shards.  On the one-GPU test box MNT753_SHARE_DEVICE=1 maps the logical devices onto the visible one (same code path
    except that the peer copy is a local copy).  Reference proofs and a synthetic set whose vectors do not divide evenly."""
    env = dict(os.environ, MNT753_SHARE_DEVICE="1")
    params, inp, expected = G.e2e_paths(curve)
    out = str(tmp_path / "proof.bin")
    env["MNT753_TRACE"] = "1"
    for flags in ([], ["--ref-order", "--unfused-h"], ["--ref-order"], ["--unfused-c"], ["--unfused-c", "--unfused-h", "--h-last"]):
        r = subprocess.run([EXE, NAME[curve], "compute", params, inp, out, "--gpus", str(n_dev)] + flags, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        assert filecmp.cmp(out, expected, shallow=False)
        # round 4: ca / cb / cc are loaded and transformed on devices 0 / 1 / 2 (0 / 1 / 0 with two devices), not all on device 0
        if "--unfused-h" not in flags:
            assert f"compute_H over devices 0 / 1 / {2 if n_dev > 2 else 0}" in r.stderr, r.stderr[-800:]
        if "--unfused-c" not in flags:
            assert f"{n_dev} devices" in r.stderr and "one MSM over H | L | B1" in r.stderr, r.stderr[-800:]
    p2, i2 = str(tmp_path / "params"), str(tmp_path / "input")
    synth_files.write_files(gpu, curve, 11 - curve, p2, i2)
    o1, oN = str(tmp_path / "one.bin"), str(tmp_path / "many.bin")
    assert subprocess.run([EXE, NAME[curve], "compute", p2, i2, o1], capture_output=True, text=True).returncode == 0
    r = subprocess.run([EXE, NAME[curve], "compute", p2, i2, oN, "--gpus", str(n_dev), "--repeat", "2"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert filecmp.cmp(o1, oN, shallow=False)


def test_more_devices_than_visible_is_refused(gpu, tmp_path):
    params, inp, _ = G.e2e_paths(0)
    env = {k: v for k, v in os.environ.items() if k != "MNT753_SHARE_DEVICE"}
    r = subprocess.run([EXE, "MNT4753", "compute", params, inp, str(tmp_path / "o"), "--gpus", "16"], capture_output=True, text=True, env=env)
    import torch
    if torch.cuda.device_count() < 16:
        assert r.returncode == 1 and "more devices requested than visible" in r.stderr


def test_params_header_is_validated(gpu, tmp_path):
    params, inp, _ = G.e2e_paths(0)
    raw = bytearray(open(params, "rb").read())
    raw[8:16] = (1 << 40).to_bytes(8, "little")          # m = 2^40
    bad = tmp_path / "params_bad"; bad.write_bytes(raw)
    r = subprocess.run([EXE, "MNT4753", "compute", str(bad), inp, str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode == 1 and ("bad params header" in r.stderr or "does not match" in r.stderr)
    trunc = tmp_path / "params_trunc"; trunc.write_bytes(bytes(raw[:-8]))
    r = subprocess.run([EXE, "MNT4753", "compute", str(trunc), inp, str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode == 1


def test_cli_errors(gpu, tmp_path):
    r = subprocess.run([EXE, "MNT4753", "compute", "/nonexistent", "/nonexistent", str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open" in r.stderr
    r = subprocess.run([EXE, "BN128", "compute", "a", "b", "c"], capture_output=True, text=True)
    assert r.returncode == 2


def test_load_file_to_device_roundtrip(gpu, tmp_path):
    """mnt753_load_file_to_device (the input loader of B::read_input): arbitrary offset, a size that is not a multiple
    of the 16 MiB staging chunk, an empty region, and the error paths (missing file, short file)."""
    rng = np.random.default_rng(5)
    data = rng.integers(0, 1 << 63, size=(40 << 20) // 8 + 3, dtype=np.uint64)   # 40 MiB + 24 B: three staging chunks
    path = tmp_path / "blob.bin"
    data.tofile(path)
    buf = gpu.DeviceBuffer.from_file(path, 0, data.nbytes)
    assert np.array_equal(buf.to_numpy(), data)
    part = gpu.DeviceBuffer.from_file(path, 8 * 1000, 8 * 123457)
    assert np.array_equal(part.to_numpy(), data[1000:1000 + 123457])
    gpu.DeviceBuffer.from_file(path, 16, 0).close()
    with pytest.raises(gpu.api.Mnt753Error):
        gpu.DeviceBuffer.from_file(tmp_path / "missing.bin", 0, 64)
    with pytest.raises(gpu.api.Mnt753Error):
        gpu.DeviceBuffer.from_file(path, data.nbytes - 8, 64)
    buf.close(); part.close()
