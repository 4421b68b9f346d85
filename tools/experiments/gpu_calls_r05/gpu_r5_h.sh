#!/bin/bash
# round 5, GPU call H: (1) the direct libff checks of the G2 / MNT6753 MSMs (new tests, the 2^20 G2 one opted in);
# (2) a timeline of the MNT6753 prove (kernel trace with timestamps, the last of four proofs) for verdict item 8
mkdir -p gpurun_out/r5h
export TMPDIR=/tmp
O=$PWD/gpurun_out/r5h
R=$PWD
( MNT753_LIBFF_G2_FULL=1 timeout 3000 python -m pytest tests/test_msm_gpu.py -m gpu -q -x -k "libff" --durations=6 ) > $O/pytest_libff.log 2>&1
echo "pytest libff rc=$?"; tail -12 $O/pytest_libff.log | cut -c1-200
K=/tmp/prove_keep; mkdir -p $K
python3 tools/synth_files.py MNT6753 15 $K/p6 $K/i6 > /dev/null
M=$R/snark-challenge-prover-reference_amd/main_hip
(cd /tmp && LD_LIBRARY_PATH=$R/snark-challenge-prover-reference_amd timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt6 -o t -- $M MNT6753 compute $K/p6 $K/i6 $K/o6 --repeat 4 > $O/prove6_traced.log 2>&1)
tail -8 $O/prove6_traced.log
python3 - <<'PY' > $O/mnt6753_prove_timeline.txt
import csv, glob
f = glob.glob("/tmp/kt6/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows: r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# the last proof: kernels after the last gap > 2 ms... simpler: take the last 1/4 by finding k_scalar_digits groups
t_end = rows[-1]["e"]
# find start of last proof = first kernel after the largest idle gap in the last 40 ms
last = [r for r in rows if r["s"] > t_end - 40_000_000]
gaps = [(last[i + 1]["s"] - max(x["e"] for x in last[:i + 1]), i) for i in range(len(last) - 1)]
g, i = max(gaps)
proof = last[i + 1:]
t0 = proof[0]["s"]
print(f"# last proof: {len(proof)} dispatches, {(proof[-1]['e'] - t0) / 1e6:.3f} ms from first kernel start to last kernel end (gap in front {g / 1e6:.3f} ms)")
print(f"# {'start_ms':>9s} {'dur_us':>9s} {'queue':>6s} {'grid':>9s}  kernel")
for r in proof:
    n = r["Kernel_Name"].split("(")[0].replace("void mnt753::", "").replace("mnt753::", "")[:80]
    print(f"  {(r['s'] - t0) / 1e6:9.3f} {(r['e'] - r['s']) / 1e3:9.1f} {r.get('Queue_Id', '?'):>6s} {int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):9d}  {n}")
PY
head -5 $O/mnt6753_prove_timeline.txt; wc -l $O/mnt6753_prove_timeline.txt
