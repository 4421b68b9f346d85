#!/usr/bin/env python3
"""Walk ONE pass of a loop body of a gfx950 kernel along its common path and count what executes (no GPU needed).

    tools/isa_walk.py <object.o> <kernel substring> <first> <last> [--policy idx=t|f,...] [--verbose]

Scalar control flow is emulated (s_mov / s_add / s_cmp on immediates, s_cbranch_scc*, and vcc built from uniform masks); a branch
whose condition comes from per-lane data (s_cbranch_execz / execnz, vcc from v_cmp) is decided by --policy (t = taken) -- the walk
stops at the first undecided one and prints its context, so the policy is built up site by site ("all lanes add an ordinary pair, a
next slot exists").  The walk ends when it leaves [first, last] or returns to `first`.  Output: executed instructions by class and the
non-multiply VALU opcodes -- the static stand-in for SQ_INSTS_VALU per slot.
"""
import collections, re, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_mix import classify, disassemble


def kernel_instrs(obj, pat):
    text = disassemble(obj)
    for f in re.split(r"\n(?=[0-9a-f]{16} <)", text):
        head = f.split("\n", 1)[0]
        if pat not in head:
            continue
        ins = []
        for ln in f.split("\n")[1:]:
            m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", ln)
            if m:
                ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
        return ins
    raise SystemExit("kernel not found")


def sreg(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"s\[(\d+):(\d+)\]$", tok)
    if m:
        return ("s", int(m.group(1)))
    m = re.match(r"s(\d+)$", tok)
    if m:
        return ("s", int(m.group(1)))
    if tok in ("vcc", "exec", "scc"):
        return (tok, 0)
    return None


def imm(tok):
    tok = tok.strip().rstrip(",")
    try:
        return int(tok, 0)
    except ValueError:
        return None


def main():
    obj, pat, a, b = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    policy = {}
    if "--policy" in sys.argv:
        for kv in sys.argv[sys.argv.index("--policy") + 1].split(","):
            if kv:
                k, v = kv.split("=")
                policy[int(k)] = v == "t"
    verbose = "--verbose" in sys.argv
    ins = kernel_instrs(obj, pat)
    addr2idx = {ad: i for i, (ad, _, _) in enumerate(ins)}
    S = {}            # known scalar values (32-bit registers; 64-bit masks under their first register: -1 / 0)
    scc = None
    vcc = None
    execz = None      # True: exec is known to be zero (s_and_saveexec with a zero mask)
    counts = collections.Counter()
    ops = collections.Counter()
    i = a
    steps = 0
    trace = []
    moves = []
    seen_auto = set()
    M64 = 0xffffffffffffffff
    while a <= i <= b and steps < 60000:
        ad, op, args = ins[i]
        steps += 1
        toks = [t.strip() for t in args.split(",")]
        cls = classify(op)
        counts[cls] += 1
        if cls in ("valu", "agpr"):
            ops[op] += 1
            if "--moves" in sys.argv and (op.startswith("v_mov") or op.startswith("v_accvgpr")):
                moves.append(i)
        nxt = i + 1
        if "saveexec" not in op and toks and toks[0] == "exec":
            execz = None

        def val(t):
            v = imm(t)
            if v is not None:
                return v
            r = sreg(t)
            if r and r[0] == "s":
                return S.get(r[1])
            if r and r[0] == "vcc":
                return vcc
            if r and r[0] == "exec":
                return -1
            return None

        def setreg(t, v):
            nonlocal vcc
            r = sreg(t)
            if r and r[0] == "s":
                S[r[1]] = v
            elif r and r[0] == "vcc":
                vcc = v

        if op in ("s_mov_b32", "s_mov_b64"):
            setreg(toks[0], val(toks[1]))
        elif op in ("s_add_i32", "s_add_u32"):
            x, y = val(toks[1]), val(toks[2])
            setreg(toks[0], None if x is None or y is None else x + y)
            scc = None
        elif op.startswith("s_cmp_"):
            x, y = val(toks[0]), val(toks[1])
            if x is None or y is None:
                scc = None
            else:
                k = op.split("_")[2]
                scc = {"eq": x == y, "lg": x != y, "lt": x < y, "gt": x > y, "le": x <= y, "ge": x >= y}[k]
        elif op in ("s_andn2_b64", "s_and_b64", "s_or_b64", "s_xor_b64", "s_orn2_b64"):
            x, y = val(toks[1]), val(toks[2])
            res = None
            if x is not None and y is not None:
                x &= M64
                y &= M64
                res = {"s_andn2_b64": x & ~y, "s_and_b64": x & y, "s_or_b64": x | y, "s_xor_b64": x ^ y, "s_orn2_b64": x | ~y}[op] & M64
                res = -1 if res == M64 else res
            elif op == "s_and_b64" and (x == 0 or y == 0):
                res = 0
            elif op == "s_andn2_b64" and (x == 0 or y == -1):
                res = 0
            elif op == "s_or_b64" and (x == -1 or y == -1):
                res = -1
            setreg(toks[0], res)
            scc = None if res is None else res != 0
        elif op == "s_cselect_b64" or op == "s_cselect_b32":
            setreg(toks[0], None if scc is None else (val(toks[1]) if scc else val(toks[2])))
        elif "saveexec" in op:
            y = val(toks[1])
            execz = None
            if op.startswith("s_and_saveexec") and y == 0:
                execz = True            # exec & 0
            setreg(toks[0], None)
            scc = None
        elif op.startswith("v_cmp") or op.startswith("v_readlane") or op.startswith("v_readfirstlane") or (op.startswith("s_") and not op.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_setprio"))):
            if toks and toks[0]:
                setreg(toks[0], None)
            if op.startswith("v_cmp") and "_e32" in op:
                vcc = None
            if op.startswith("s_") and not op.startswith("s_load"):
                scc = None
        if op.startswith("s_cbranch") or op == "s_branch":
            simm = int(toks[0])
            if simm >= 32768:
                simm -= 65536
            tgt = addr2idx.get(ad + 4 + simm * 4)
            cond = None
            if op == "s_branch":
                cond = True
            elif op == "s_cbranch_scc0":
                cond = None if scc is None else (not scc)
            elif op == "s_cbranch_scc1":
                cond = scc
            elif op == "s_cbranch_vccz":
                cond = None if vcc is None else (vcc == 0)
            elif op == "s_cbranch_vccnz":
                cond = None if vcc is None else (vcc != 0)
            elif op == "s_cbranch_execz" and execz:
                cond = True
            elif op == "s_cbranch_execnz" and execz:
                cond = False
            if cond is None:
                if i in policy:
                    cond = policy[i]
                elif "--auto" in sys.argv:
                    # default guess: an execz branch falls into its region, an execnz branch is taken; every guess is listed for review
                    cond = op == "s_cbranch_execnz" or op == "s_cbranch_vccnz"
                    if i not in seen_auto:
                        seen_auto.add(i)
                        print(f"  auto: {i} {op} -> {tgt} {'taken' if cond else 'not taken'}   | " + " ; ".join(ins[k][1] + " " + ins[k][2][:34] for k in range(max(0, i - 3), i)))
                else:
                    print(f"undecided branch at {i}: {op} -> {tgt}   (add --policy {i}=t or {i}=f)")
                    for k in range(max(0, i - 8), i + 1):
                        print("     ", k, ins[k][1], ins[k][2][:70])
                    break
            if verbose:
                trace.append((i, op, tgt, cond))
            if cond:
                if tgt is None:
                    break
                if tgt == a and steps > 1:
                    i = tgt
                    break
                nxt = tgt
        i = nxt
    print("executed:", dict(counts), " VALU total", counts["mad64"] + counts["valu"] + counts["agpr"], " stopped at", i)
    print("non-MAD VALU by opcode:", ops.most_common(30))
    if moves:
        runs = []
        for m in moves:
            if runs and m - runs[-1][1] <= 3:
                runs[-1][1] = m; runs[-1][2] += 1
            else:
                runs.append([m, m, 1])
        print("executed register moves (first, last, count):", [tuple(r) for r in runs if r[2] >= 4])
    if verbose:
        for t in trace:
            print("   ", t)


main()
