#!/usr/bin/env python3
"""GPU box: the kernels of ONE resident proof on a time axis -- where the device idles inside the timing window.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- <main_hip ... --repeat 3>
    python3 tools/prove_timeline.py /tmp/tl [--gap-ms 1.0] [--all]

Dispatches are split into proofs at pauses of the device longer than --gap-ms; the LAST group is printed: per kernel start (ms from the
first), duration, queue; then the span, the union of busy intervals, the idle intervals above 20 us and the time per kernel name."""
import argparse
import csv
import glob
import os
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("mnt753::", "")
    return n[:64]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--gap-ms", type=float, default=1.0)
    ap.add_argument("--all", action="store_true", help="print every dispatch of the group")
    ap.add_argument("--group", type=int, default=-1)
    a = ap.parse_args()
    files = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit("no *kernel_trace.csv under " + a.dir)
    rows = []
    for r in csv.DictReader(open(files[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    rows.sort()
    groups, cur, end = [], [], None
    for r in rows:
        if end is not None and r[0] - end > a.gap_ms * 1e6:
            groups.append(cur); cur = []
        cur.append(r); end = max(end or 0, r[1])
    groups.append(cur)
    print(f"{len(rows)} dispatches, {len(groups)} groups at gaps > {a.gap_ms} ms; sizes {[len(g) for g in groups]}")
    g = groups[a.group]
    t0 = g[0][0]
    span = (max(r[1] for r in g) - t0) / 1e6
    busy, idle, cur_end = 0.0, [], g[0][0]
    for s, e, n, q, st in g:
        if s > cur_end:
            if s - cur_end > 20e3:
                idle.append(((cur_end - t0) / 1e6, (s - cur_end) / 1e6, n))
            busy += 0
            cur_end = s
        if e > cur_end:
            busy += (e - max(s, cur_end)) / 1e6
            cur_end = e
    print(f"group {a.group}: {len(g)} dispatches, span {span:.3f} ms, device busy (union) {busy:.3f} ms, idle {span - busy:.3f} ms")
    print("idle intervals > 20 us (start ms, length ms, kernel that ended it):")
    for s, l, n in idle:
        print(f"   {s:8.3f}  {l:7.3f}  {short(n)}")
    by = {}
    for s, e, n, q, st in g:
        k = short(n); by.setdefault(k, [0, 0.0]); by[k][0] += 1; by[k][1] += (e - s) / 1e6
    print("time per kernel (sum of durations, overlapping ones both counted):")
    for k, (c, t) in sorted(by.items(), key=lambda x: -x[1][1])[:28]:
        print(f"   {k:66s} {c:4d} {t:8.3f}")
    if a.all:
        for s, e, n, q, st in g:
            print(f"   {(s - t0) / 1e6:8.3f} {(e - s) / 1e6:7.3f}  q{q} s{st}  {short(n)}")


if __name__ == "__main__":
    main()
