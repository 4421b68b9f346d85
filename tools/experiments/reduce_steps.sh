# GPU box: per-dispatch durations of k_reduce_step of one MSM (rocprofv3 --kernel-trace), in launch order
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/rs -o x -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prove --no-extras > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/rs/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
red = [(r["Kernel_Name"].split("(")[0][-40:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_reduce_step" in r["Kernel_Name"] or "k_reduce_collect" in r["Kernel_Name"]]
last = red[-20:]
t0 = last[0][2]
print("step  dur_us  start_us  gap_before_us")
prev_end = None
for i, (n, d, s, e) in enumerate(last):
    print(f"{i:3d} {d:8.1f} {(s - t0) / 1e3:9.1f} {((s - prev_end) / 1e3 if prev_end else 0):8.1f}  {n}")
    prev_end = e
print("sum of durations %.1f us, span %.1f us" % (sum(x[1] for x in last), (last[-1][3] - t0) / 1e3))
PY
