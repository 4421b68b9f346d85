#!/bin/sh
# GPU box: the evidence bundle of a round -> gpurun_out/prof/ (copy what is to be judged into profiles/rNN/).
#   bench line, rocprofv3 kernel stats of the bench command, the two PMC passes (separate runs, as the microarch
#   guide prescribes), kernel stats of a G2 MSM, and the full-size proves.  usage: tools/collect_profiles.sh
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/prof; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_stderr.log
# round 3: the slice sweep behind pair_levels() and the predicted 1/2/4/8-GPU curve, board power / clocks under load, the compiler
# reproducers, and the N > 1 flow of bench.py on the shared device (gloo)
python3 tools/slice_sweep.py --out $O/slice_sweep.json > $O/slice_sweep.log 2>&1
sh tools/experiments/power_sample.sh > $O/power_and_clocks.txt 2>&1
sh tools/compiler_repro/check.sh > $O/compiler_repro.log 2>&1
BENCH_SHARE_GPU=1 python3 bench.py --gpus 4 --steps 5 --warmup 2 > $O/bench_4ranks_shared_gpu.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
# exactly the timed loop of the driver's command (20 steps + 5 warm-up of the table-mode 2^20 G1 MSM, nothing else): the per-kernel
# averages of kt_loop are what `roofline.kernel_ms` and the per-kernel table of DESIGN.md 4.3 must agree with
rocprofv3 --kernel-trace --stats -d $O/kt_loop -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prove --no-extras --no-traffic > $O/bench_loop_under_rocprof.json 2>/dev/null
# the other legs (table-less MSM, FFT / compute_H, G2, slice sweep) in a trace of their own
rocprofv3 --kernel-trace --stats -d $O/kt_extras -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prove --no-traffic > $O/bench_extras_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prove --no-extras --no-traffic > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prove --no-extras --no-traffic > /dev/null 2>&1
CURVE=0 GROUP=2 rocprofv3 --kernel-trace --stats -d $O/kt_g2 -o g2 -- python3 $R/tools/dev_msm_big.py 20 3 > $O/g2_msm_2p20.log 2>/dev/null
CURVE=1 GROUP=2 rocprofv3 --kernel-trace --stats -d $O/kt_g2m6 -o g2 -- python3 $R/tools/dev_msm_big.py 15 3 > $O/g2_mnt6_msm_2p15.log 2>/dev/null
cd $R
export TMPDIR=/tmp
K=/tmp/prove_keep; mkdir -p $K
python3 tools/synth_files.py MNT4753 20 $K/p4 $K/i4 > /dev/null
python3 tools/synth_files.py MNT6753 15 $K/p6 $K/i6 > /dev/null
M=$R/snark-challenge-prover-reference_amd/main_hip
{ echo "== main_hip MNT4753 d = 2^20 - 1, three proofs against resident parameters"; $M MNT4753 compute $K/p4 $K/i4 $K/o4 --repeat 3; sha256sum $K/o4;
  echo "== the reference's call order and its unfused compute_H (--ref-order --unfused-h)"; $M MNT4753 compute $K/p4 $K/i4 $K/o4r --ref-order --unfused-h | grep -i "total\|load"; sha256sum $K/o4r;
  echo "== five separate MSMs (--unfused-c) instead of three"; $M MNT4753 compute $K/p4 $K/i4 $K/o4u --unfused-c --repeat 2 | grep -i "total\|load"; sha256sum $K/o4u;
  echo "== two logical devices sharing the one GPU of this box (MNT753_SHARE_DEVICE=1 --gpus 2)"; MNT753_SHARE_DEVICE=1 $M MNT4753 compute $K/p4 $K/i4 $K/o4s --gpus 2 --repeat 2 | grep -i "total\|load"; sha256sum $K/o4s;
  echo "== eight logical devices sharing the one GPU (MNT753_SHARE_DEVICE=1 --gpus 8): every code path of the 8-way split, none of its speed"; MNT753_SHARE_DEVICE=1 $M MNT4753 compute $K/p4 $K/i4 $K/o4e --gpus 8 --repeat 2 | grep -i "total\|load"; sha256sum $K/o4e;
  echo "== reference-minted hashes (tests/golden/oracle_hashes.json)"; grep output_sha256 tests/golden/oracle_hashes.json; } > $O/full_prove_MNT4753_2p20.log 2>&1
{ echo "== main_hip MNT6753 d = 2^15 - 1"; $M MNT6753 compute $K/p6 $K/i6 $K/o6 --repeat 3; sha256sum $K/o6;
  echo "== CPU: the reference prover (oracle/_ref/main, $(nproc) hardware threads)"; if [ -x oracle/_ref/main ]; then oracle/_ref/main MNT6753 compute $K/p6 $K/i6 $K/o6ref 2>&1 | grep -i "total time"; sha256sum $K/o6ref; fi;
  echo "== CPU: the oracle restatement (oracle_main)"; oracle/oracle_main MNT6753 compute $K/p6 $K/i6 $K/o6or; sha256sum $K/o6or; } > $O/full_prove_MNT6753_2p15.log 2>&1
# round 6: the one-shot prover (one job: no window tables, no level buffers, no warm-up; host/main.cpp) -- the wall clock of the whole
# process, three times with pauses, against --tables; and the product's self-test per curve
wall() { t0=$(date +%s.%N); "$@" > $K/stdout.txt 2> $K/stderr.txt; rc=$?; t1=$(date +%s.%N); echo "rc $rc wall $(python3 -c "print(round($t1 - $t0, 3))") s | $(grep -E 'load params:|Total time from' $K/stdout.txt | tr '\n' ' ')"; }
{ for k in 1 2 3; do sleep 20; echo -n "one-shot : "; wall $M MNT4753 compute $K/p4 $K/i4 $K/o1; sleep 20; echo -n "--tables : "; wall $M MNT4753 compute $K/p4 $K/i4 $K/o2 --tables; cmp $K/o1 $K/o2 && echo "same bytes"; done;
  sleep 20; MNT753_TRACE_LOAD=1 $M MNT4753 compute $K/p4 $K/i4 $K/o1 2>&1 | grep -E "load params|one-shot|Total"; sha256sum $K/o1;
  echo "== MNT6753 2^15, one-shot"; sleep 5; wall $M MNT6753 compute $K/p6 $K/i6 $K/o61; sha256sum $K/o61; $M MNT4753 self-test; } > $O/one_shot_wall.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt_prove -o prove -- $M MNT4753 compute $K/p4 $K/i4 $K/o4 --repeat 2 > $O/prove_under_rocprof.log 2>&1
cd $R; rm -rf $K
# keep the summaries, drop the bulky databases
python3 - <<'PY'
import sqlite3, glob, os, csv, collections
O = os.path.join(os.getcwd(), "gpurun_out", "prof")
def tables(cur): return [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
for db in glob.glob(O + "/*/*_results.db"):
    con = sqlite3.connect(db); cur = con.cursor(); t = tables(cur)
    kd = [x for x in t if "kernel_dispatch" in x][0]; ks = [x for x in t if "kernel_symbol" in x][0]
    tag = os.path.basename(os.path.dirname(db))
    rows = list(cur.execute(f"select s.display_name, d.end - d.start, d.id from {kd} d join {ks} s on d.kernel_id = s.id"))
    agg = collections.defaultdict(list)
    for n, dt, _ in rows: agg[n].append(dt)
    total = sum(sum(v) for v in agg.values()) or 1
    with open(f"{O}/{tag}_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([n, len(v), sum(v), sum(v) / len(v), round(100 * sum(v) / total, 2), min(v), max(v)])
    pm = [x for x in t if "pmc_event" in x]; pi = [x for x in t if "info_pmc" in x]
    if pm and pi:
        q = f"select s.display_name, p.symbol, e.value from {pm[0]} e join {pi[0]} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id"
        try:
            pa = collections.defaultdict(list)
            for n, sym, val in cur.execute(q): pa[(n, sym)].append(val)
            if pa:
                with open(f"{O}/{tag}_pmc.csv", "w", newline="") as f:
                    w = csv.writer(f); w.writerow(["kernel", "counter", "launches", "avg_value"])
                    for (n, sym), v in sorted(pa.items(), key=lambda kv: -sum(kv[1])): w.writerow([n, sym, len(v), sum(v) / len(v)])
        except Exception as ex:
            open(f"{O}/{tag}_pmc_error.txt", "w").write(repr(ex))
    con.close(); os.remove(db)
PY
python3 - <<'PY'
# HBM traffic of the bucket-accumulation phase of one G1 MSM: FETCH_SIZE / WRITE_SIZE (KB) summed over the kernels of the phase
import csv, json, os
O = os.path.join(os.getcwd(), "gpurun_out", "prof")
def per_msm(f, ctr):
    tot = 0.0; launches = {}
    for r in csv.DictReader(open(f)):
        n = r["kernel"]
        if r["counter"] != ctr: continue
        if "k_pair_level<mnt753::Mnt4G1" in n or "k_bucket_accumulate<mnt753::Mnt4G1" in n or "k_pair_fix" in n:
            launches[n.split("(")[0]] = (int(r["launches"]), float(r["avg_value"]))
    acc = [v for k, v in launches.items() if "k_bucket_accumulate" in k]
    msms = acc[0][0] if acc else 1
    for k, (cnt, avg) in launches.items(): tot += cnt * avg / msms
    return tot, launches, msms
try:
    f, lf, m = per_msm(O + "/pmc_fetch_pmc.csv", "FETCH_SIZE"); w, lw, _ = per_msm(O + "/pmc_write_pmc.csv", "WRITE_SIZE")
    import hashlib
    h = hashlib.sha256()
    for name in ("msm_kernels.hip.h", "msm_host.hpp", "curve753.hip.h", "fp753.hip.h"):
        h.update(open(os.path.join(os.getcwd(), "snark-challenge-prover-reference_amd", "csrc", name), "rb").read())
    json.dump({"phase": "bucket accumulation of one 2^20 G1 MSM: k_pair_level x levels + k_bucket_accumulate (+ k_pair_fix)",
               "kernels_fingerprint": h.hexdigest()[:16],
               "msms_averaged": m, "FETCH_SIZE_KB_per_msm": f, "WRITE_SIZE_KB_per_msm": w,
               "raw_bytes_per_msm": (f + w) * 1024, "hbm_bytes_per_launch": (2 * f + w) * 1024,
               "per_kernel_FETCH_KB": {k: {"launches": c, "avg": a} for k, (c, a) in lf.items()},
               "per_kernel_WRITE_KB": {k: {"launches": c, "avg": a} for k, (c, a) in lw.items()},
               "correction": "MI355X_MICROARCH.md HBM section: counters in KB; FETCH_SIZE doubled for 16-B-per-lane loads on gfx950 (row gathers: uncalibrated pattern, raw figure kept beside it)",
               "algorithmic_bytes_per_launch": 301989888}, open(O + "/accumulate_traffic.json", "w"), indent=1)
except Exception as ex:
    open(O + "/accumulate_traffic_error.txt", "w").write(repr(ex))
PY
ls -la $O
