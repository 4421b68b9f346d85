#!/usr/bin/env python3
"""Which `__global__` x template instantiations of libmnt753_hip.so does the GPU suite actually launch?

hipcc has miscompiled these kernels three times (DESIGN.md 4.2 findings 1 and 6, 4.9), each time in ONE instantiation, and the last time
it was caught by a test written for another reason.  This script lists every kernel symbol of the product library's gfx950 code object
and compares it with the kernel names of a `rocprofv3 --kernel-trace` of the GPU suite:

    # GPU box
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/kcov -o k -- python3 -m pytest $REPO/tests -m gpu -q --deselect ...
    python3 tools/kernel_coverage.py --traces /tmp/kcov --out profiles/r05/kernel_coverage.txt

    # build container (no GPU): only the list of instantiations
    python3 tools/kernel_coverage.py --list

Output: per kernel symbol (demangled) the number of launches over the whole suite and the largest grid it was launched with (a launch
with one workgroup says little about a kernel whose bugs showed at depth), then the symbols that were NEVER launched."""
import argparse
import collections
import csv
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_symbols(lib):
    """demangled names of the kernels (symbols with a .kd kernel descriptor) of the gfx950 code object(s) inside `lib`"""
    d = tempfile.mkdtemp()
    x = os.path.join(d, "lib.so")
    subprocess.check_call(["cp", lib, x])
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", x], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    names = set()
    for co in glob.glob(os.path.join(d, "*amdgcn*")):
        out = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--symbols", "--wide", co], text=True)
        for line in out.splitlines():
            parts = line.split()
            if parts and parts[-1].endswith(".kd"):
                names.add(parts[-1][:-3])
    subprocess.call(["rm", "-rf", d])
    if not names:
        return []
    dem = subprocess.check_output(["c++filt"], input="\n".join(sorted(names)), text=True).splitlines()
    return sorted(set(normalise(n) for n in dem))


def normalise(name):
    """`void mnt753::k_x<mnt753::Mnt4G1, true>(args...)` -> `k_x<Mnt4G1, true>`"""
    n = name.strip()
    n = re.sub(r"\s*\[clone .*\]$", "", n)
    n = n.replace("(anonymous namespace)::", "").replace("mnt753::", "")
    if n.startswith("void "):
        n = n[5:]
    # cut the argument list: the last top-level '(' that closes at the end
    depth, cut = 0, None
    for i, ch in enumerate(n):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return (n[:cut] if cut is not None else n).strip()


def launched(trace_dir):
    """kernel name -> [launches, largest grid (workgroups)] over every *_kernel_trace.csv under trace_dir (one per traced process)"""
    acc = collections.defaultdict(lambda: [0, 0])
    files = glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = normalise(row.get("Kernel_Name") or row.get("kernel_name") or "")
                if not name:
                    continue
                try:
                    grid = int(row.get("Grid_Size_X", row.get("grid_size_x", 0))) * int(row.get("Grid_Size_Y", row.get("grid_size_y", 1)) or 1)
                    wg = int(row.get("Workgroup_Size_X", row.get("workgroup_size_x", 1)) or 1) * int(row.get("Workgroup_Size_Y", row.get("workgroup_size_y", 1)) or 1)
                    groups = grid // max(wg, 1)
                except (TypeError, ValueError):
                    groups = 0
                a = acc[name]
                a[0] += 1
                a[1] = max(a[1], groups)
    return acc, len(files)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "snark-challenge-prover-reference_amd", "libmnt753_hip.so"))
    ap.add_argument("--test-lib", default=os.path.join(ROOT, "snark-challenge-prover-reference_amd", "libmnt753_hip_test.so"))
    ap.add_argument("--traces", help="directory of a rocprofv3 --kernel-trace --output-format csv run of the GPU suite")
    ap.add_argument("--list", action="store_true")
    ap.add_argument("--out")
    args = ap.parse_args()
    prod = kernel_symbols(args.lib)
    lines = [f"# {len(prod)} kernel instantiations in {os.path.basename(args.lib)} (gfx950 code object)"]
    if args.list or not args.traces:
        lines += prod
        text = "\n".join(lines) + "\n"
    else:
        acc, n_files = launched(args.traces)
        lines.append(f"# launches over the GPU suite: {n_files} traced processes, {sum(v[0] for v in acc.values())} dispatches")
        lines.append(f"# {'kernel':100s} launches  largest grid (workgroups)")
        never = []
        for k in prod:
            if k in acc:
                lines.append(f"{k:102s} {acc[k][0]:8d}  {acc[k][1]:8d}")
            else:
                never.append(k)
        lines.append("")
        lines.append(f"# NEVER launched by the suite: {len(never)} of {len(prod)}")
        lines += never
        test = set(kernel_symbols(args.test_lib)) if os.path.exists(args.test_lib) else set()
        other = sorted(k for k in acc if k not in set(prod))
        lines.append("")
        lines.append(f"# launched but not a symbol of the product library ({len(other)}): the test library's hooks, torch, RCCL, the HIP runtime's copy / fill kernels")
        lines += [f"{k:102s} {acc[k][0]:8d}  {acc[k][1]:8d}  {'(test library)' if k in test else ''}" for k in other]
        text = "\n".join(lines) + "\n"
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
