#!/bin/bash
# round 4, GPU call W: timeline of the MNT4753 2^20 prove per queue: when do the throughput phases of the three MSMs run, what is exposed
mkdir -p gpurun_out/r4w
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4w
R=$PWD
K=/tmp/pk; mkdir -p $K
python3 tools/synth_files.py MNT4753 20 $K/p4 $K/i4 > /dev/null
M=$R/snark-challenge-prover-reference_amd/main_hip
cd /tmp
rocprofv3 --kernel-trace -d $O/kt -o t -- $M MNT4753 compute $K/p4 $K/i4 $K/o4 --repeat 2 > $O/prove.log 2>&1
cd $R
grep -i "total time\|gpu:" $O/prove.log | tail -4
python3 - <<'PY'
import sqlite3, glob, os
O = os.path.join(os.getcwd(), "gpurun_out", "r4w")
for db in glob.glob(f"{O}/kt/**/*_results.db", recursive=True):
    con = sqlite3.connect(db); cur = con.cursor()
    t = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in t if "kernel_dispatch" in x][0]; ks = [x for x in t if "kernel_symbol" in x][0]
    rows = list(cur.execute(f"select s.display_name, d.start, d.end, d.queue_id from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    # the last proof: kernels after the last gap > 20 ms ... use the last k_r1cs-free window: find last gap > 1 ms preceded by >100 ms of work
    cut = 0; busy_end = rows[0][2]
    for i in range(1, len(rows)):
        if rows[i][1] - busy_end > 900_000: cut = i
        busy_end = max(busy_end, rows[i][2])
    rows = rows[cut:]
    t0 = rows[0][1]
    with open(f"{O}/mnt4753_prove_timeline.txt", "w") as f:
        for n, st, en, q in rows:
            if en - st > 250_000 or "edge" in n or "reduce_collect" in n or "points_to_wire" in n:
                f.write(f"at {(st - t0) / 1e6:9.3f} ms  {(en - st) / 1e3:9.1f} us  q{q}  {n.split('(')[0][:80]}\n")
    print(open(f"{O}/mnt4753_prove_timeline.txt").read())
    con.close(); os.remove(db)
PY
