"""GPU: the N > 1 flow of bench.py -- the north-star split of ONE array into contiguous slices, one rank per slice, all_gather of
one projective point per rank, serial fold (multiexp.tcc:417-440) -- on the one-GPU test box: BENCH_SHARE_GPU=1 lets the ranks share
the device and exchange over gloo, everything else (slicing, the strong headline, the weak object, parity through the discrete logs,
max-over-ranks timing, the JSON contract) is the code the driver's multi-GPU run executes over RCCL."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 4])
def test_bench_multi_rank_flow(gpu, world):
    env = dict(os.environ, BENCH_SHARE_GPU="1")
    # --prove-log2-d 14 10: the prove legs at the reference generator's `fast` sizes, for which tests/golden/oracle_hashes.json holds the
    # reference's proofs (the driver's run uses the metric's own 2^20 / 2^15)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--log-n", "13",
                        "--prove-log2-d", "14", "10"], capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == world and j["steps"] == 2 and j["warmup"] == 1
    assert j["scaling"] == "strong" and j["parity_ok"] is True
    assert j["config"]["points"] == 1 << 13 and j["config"]["points_per_gpu"] in ((1 << 13) // world, (1 << 13) - ((1 << 13) // world) * (world - 1))
    assert j["weak"]["scaling"] == "weak" and j["weak"]["parity_ok"] is True
    assert j["value"] > 0 and j["weak"]["value"] > 0 and j["unit"] == "points/s"
    for key in ("metric", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "roofline"):
        assert key in j
    # round 5: the prove at N GPUs is in the line -- main_hip --gpus N started by rank 0 before any rank touched a GPU (libsnark/main.cpp:203-270
    # is the window, multiexp.tcc:417-440 the sharding), its bytes are the reference's, also from a cold process and with the fold over RCCL requested
    minted = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_hashes.json")))
    for key, name in (("prove", "MNT4753_2p14"), ("prove_mnt6753", "MNT6753_2p10")):
        pr = j[key]
        assert pr["n_gpus"] == world and pr["parity_ok"] is True, pr
        assert pr["sha256"] == minted[name]["output_sha256"] and pr["input_to_output_s"] > 0
        assert pr["cold_process"]["same_bytes"] is True and pr["cold_process"]["input_to_output_s"] > 0
        assert pr["fold_rccl"]["same_bytes"] is True and pr["fold_rccl"]["folded"].startswith("on the host")   # logical devices share the GPU here
        # round 6: the N > 1 line MEASURES what DESIGN.md section 5 assumed (2.0 ms per 100 MB over xGMI): one 100 MB mnt753_copy_peer_async per
        # copy of the sharded prove (1 -> 0, 2 -> 0, 0 -> g), timed by rank 0's main_hip --gpus N --peer-bench child, with the path the box granted.
        # Here the logical devices share ONE GPU: the fields must be there and say so.
        copies = pr["peer_copy_100MB"]
        assert {(c["src"], c["dst"]) for c in copies} >= {(1, 0)} | {(0, g) for g in range(1, world)}
        assert all(c["ms"] > 0 and c["GB_per_s"] > 0 and "same GPU" in c["path"] for c in copies)
        assert pr["peer_copy_100MB_to_device0_ms"] > 0 and "share one GPU" in pr["peer_copy_note"]
        assert pr["peer_access"]["ordered_pairs"] == world * (world - 1) and pr["peer_access"]["same_gpu"] == world * (world - 1)


def test_bench_single_gpu_contract_small(gpu):
    """The N = 1 line at a small size: every field of the contract, no prove / extras legs (those need the 2^20 workload)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--log-n", "13", "--no-cpu-baseline",
                        "--prove-log2-d", "14", "10"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 1 and j["scaling"] == "strong" and j["parity_ok"] is True   # the N = 1 point of the strong series (one 2^20 array at every N)
    assert j["prove"]["n_gpus"] == 1 and j["prove"]["parity_ok"] is True and j["prove"]["cold_process"]["same_bytes"] is True
    assert j["prove"]["cold_process"]["input_to_output_s"] > 0 and j["prove"]["input_to_output_s"] > 0
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(j["roofline"])
    # round 6: the reference's CLI is a one-proof process (libsnark/main.cpp:274-293) -- the wall clock of `main_hip <curve> compute ...` invoked
    # that way (no window tables, no warm-up MSM) is in the line beside the resident prover's, same bytes
    one = j["prove"]["one_shot"]
    assert one["same_bytes"] is True and one["one_shot_policy_applied"] is True and j["prove"]["one_shot_wall_s"] == one["wall_incl_params_s"] > 0
    assert j["roofline"]["bound"].startswith("int-mad") and j["roofline"]["modmul_peak_first_principles_per_s"] > j["roofline"]["modmul_peak_per_s"]
