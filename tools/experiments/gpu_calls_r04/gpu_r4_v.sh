#!/bin/bash
# round 4, GPU call V: timeline of the MNT6753 2^15 prove (kernel start / end per queue): where are the gaps?
mkdir -p gpurun_out/r4v
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4v
R=$PWD
K=/tmp/pk; mkdir -p $K
python3 tools/synth_files.py MNT6753 15 $K/p6 $K/i6 > /dev/null
M=$R/snark-challenge-prover-reference_amd/main_hip
cd /tmp
rocprofv3 --kernel-trace -d $O/kt -o t -- $M MNT6753 compute $K/p6 $K/i6 $K/o6 --repeat 3 > $O/prove.log 2>&1
cd $R
grep -i "total time\|gpu:" $O/prove.log | tail -4
python3 - <<'PY'
import sqlite3, glob, os
O = os.path.join(os.getcwd(), "gpurun_out", "r4v")
for db in glob.glob(f"{O}/kt/**/*_results.db", recursive=True):
    con = sqlite3.connect(db); cur = con.cursor()
    t = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in t if "kernel_dispatch" in x][0]; ks = [x for x in t if "kernel_symbol" in x][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = list(cur.execute(f"select s.display_name, d.start, d.end, {('d.' + qcol) if qcol else '0'} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    # the last proof: from the last k_r1cs / first kernel after a gap > 3 ms
    starts = [i for i in range(1, len(rows)) if rows[i][1] - max(r[2] for r in rows[max(0, i - 50):i]) > 2_000_000]
    first = starts[-1] if starts else 0
    t0 = rows[first][1]; busy_end = t0; gaps = []
    with open(f"{O}/mnt6753_prove_timeline.txt", "w") as f:
        for n, st, en, q in rows[first:]:
            gap = st - busy_end
            if gap > 20_000: gaps.append((round((st - t0) / 1e3), round(gap / 1e3)))
            busy_end = max(busy_end, en)
            f.write(f"at {(st - t0) / 1e3:9.1f} us  {(en - st) / 1e3:8.1f} us  q{q}  {n.split('(')[0][:80]}\n")
    print("kernels of the last proof:", len(rows) - first, "span", round((busy_end - t0) / 1e3), "us; gaps > 20 us with no kernel running (at, length):", gaps)
    con.close(); os.remove(db)
PY
