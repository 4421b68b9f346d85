#!/usr/bin/env python3
"""Static walk of one kernel's control flow (no GPU needed): lists every forward conditional branch inside a line range with what it
skips, so that the COMMON path of a loop body (all lanes on the ordinary case) can be counted by naming the regions that are rare.

    tools/isa_paths.py <object.o> <kernel substring> <first line> <last line> [--skip a-b,c-d ...]

Lines are 0-based instruction indices of the kernel as printed by tools/isa_mix.py.  With --skip the instruction classes of the range
are summed without the named sub-ranges (the rare paths) -- a static stand-in for SQ_INSTS_VALU per slot.
"""
import collections, re, sys
sys.path.insert(0, __import__("os").path.dirname(__file__))
from isa_mix import disassemble, classify

def kernel_instrs(obj, pat):
    text = disassemble(obj)
    for f in re.split(r"\n(?=[0-9a-f]{16} <)", text):
        head = f.split("\n", 1)[0]
        if pat not in head: continue
        ins = []
        for ln in f.split("\n")[1:]:
            m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", ln)
            if not m: continue
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
        return ins
    raise SystemExit("kernel not found")

def main():
    obj, pat, a, b = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    skips = []
    if "--skip" in sys.argv:
        for s in sys.argv[sys.argv.index("--skip") + 1].split(","):
            x, y = s.split("-"); skips.append((int(x), int(y)))
    ins = kernel_instrs(obj, pat)
    addr2idx = {ad: i for i, (ad, _, _) in enumerate(ins)}
    def cnt(lo, hi, sk=()):
        c = collections.Counter()
        for i in range(lo, hi + 1):
            if any(x <= i <= y for x, y in sk): continue
            c[classify(ins[i][1])] += 1
        return c
    if skips:
        c = cnt(a, b, skips)
        print("common path [%d, %d] minus %s:" % (a, b, skips), dict(c), " VALU total", c["mad64"] + c["valu"] + c["agpr"])
        ops = collections.Counter(ins[i][1] for i in range(a, b + 1) if not any(x <= i <= y for x, y in skips) and classify(ins[i][1]) in ("valu", "agpr"))
        print("   non-MAD VALU by opcode:", ops.most_common(25))
        return
    for i in range(a, b + 1):
        ad, op, args = ins[i]
        if op.startswith("s_cbranch") or op == "s_branch":
            m = re.search(r"<[^>]*\+0x([0-9a-f]+)>", ins[i][2] + " " )
            # target = address after the instruction + simm16 * 4; objdump prints the label offset instead: recompute
            simm = int(args.split()[0])
            if simm >= 32768: simm -= 65536
            tgt = ad + 4 + simm * 4
            j = addr2idx.get(tgt)
            if j is None: continue
            if j > i:
                c = cnt(i + 1, j - 1)
                prev = " ; ".join(ins[k][1] + " " + ins[k][2][:40] for k in range(max(a, i - 2), i))
                print(f"{i:6d} {op:18s} -> {j:6d}  skips {j - i - 1:5d}: mad {c['mad64']:5d} valu {c['valu']:5d} agpr {c['agpr']:4d} lds {c['lds']:3d} vmem {c['vmem']:3d}   | {prev}")
            else:
                print(f"{i:6d} {op:18s} -> {j:6d}  (backward)")
main()
