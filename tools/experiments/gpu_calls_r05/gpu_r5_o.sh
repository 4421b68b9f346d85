#!/bin/bash
# round 5, GPU call O: why is the FIRST proof of a process ~0.8 ms (MNT6753) / ~1.5 ms (MNT4753) longer than the ones behind it?
# kernel span (first kernel start .. last kernel end) and host gaps per proof, from a timestamped kernel trace of --repeat 4
mkdir -p gpurun_out/r5o
export TMPDIR=/tmp
O=$PWD/gpurun_out/r5o
R=$PWD
K=/tmp/prove_keep; mkdir -p $K
python3 tools/synth_files.py MNT6753 15 $K/p6 $K/i6 > /dev/null
M=$R/snark-challenge-prover-reference_amd/main_hip
(cd /tmp && LD_LIBRARY_PATH=$R/snark-challenge-prover-reference_amd timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt6 -o t -- $M MNT6753 compute $K/p6 $K/i6 $K/o6 --repeat 4 > $O/prove6_traced.log 2>&1)
grep "Total time from" $O/prove6_traced.log
python3 - <<'PY' > $O/first_proof_vs_later.txt
import csv, glob
f = glob.glob("/tmp/kt6/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows: r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"]); r["n"] = r["Kernel_Name"].split("(")[0].replace("void mnt753::", "").replace("mnt753::", "")
rows.sort(key=lambda r: r["s"])
# a proof starts at the k_scalar_digits<1> of the G2 MSM that follows an idle gap > 0.5 ms; take the last 4 such groups
starts = []
end_so_far = 0
for i, r in enumerate(rows):
    if i and r["s"] - end_so_far > 400_000: starts.append(i)
    end_so_far = max(end_so_far, r["e"])
starts = starts[-4:]
bounds = starts + [len(rows)]
print("# per proof of --repeat 4 (MNT6753 2^15): kernel span, busy time (union of kernel intervals), idle inside the span, gap to the previous proof's last kernel")
for k in range(4):
    seg = rows[bounds[k]:bounds[k + 1]]
    t0, t1 = seg[0]["s"], max(x["e"] for x in seg)
    # union of intervals
    busy = 0; cur_s, cur_e = None, None
    for x in sorted(seg, key=lambda x: x["s"]):
        if cur_e is None or x["s"] > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = x["s"], x["e"]
        else: cur_e = max(cur_e, x["e"])
    busy += cur_e - cur_s
    prev_end = max(x["e"] for x in rows[:bounds[k]])
    big = {}
    for x in seg:
        if x["e"] - x["s"] > 300_000: big[x["n"]] = big.get(x["n"], 0) + (x["e"] - x["s"])
    print(f"proof {k + 1}: {len(seg)} dispatches, span {(t1 - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms, gap in front {(t0 - prev_end) / 1e6:.3f} ms")
    print("    " + ", ".join(f"{n[:44]} {v / 1e6:.2f}" for n, v in sorted(big.items(), key=lambda kv: -kv[1])[:8]))
PY
cat $O/first_proof_vs_later.txt
