#!/usr/bin/env python3
"""Static instruction mix of the loops of one kernel in a gfx950 object (no GPU needed).

    tools/isa_mix.py build/msm_inst_mnt4g1.o 'k_pair_levelINS_6Mnt4G1ELb0ELb0'

Disassembles the device code object, finds every backward branch of the kernel (= a loop), and prints for each loop
body the number of instructions by class: 64-bit multiply-adds (the arithmetic the multiplier roof counts), other
VALU, AGPR moves, scalar ALU, LDS, vector memory, waits, branches.  Nested loops are reported innermost-first with
their own counts; an outer loop's line includes its inner loops' bodies once.  Used to judge a kernel change before
spending GPU minutes on it: the share of v_mad_u64_u32 in the hot loop is what `modmul_frac` can reach at best.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disassemble(obj):
    d = tempfile.mkdtemp()
    x = os.path.join(d, "x.o")
    subprocess.check_call(["cp", obj, x])
    subprocess.run([OBJDUMP, "--offloading", x], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    co = [f for f in os.listdir(d) if "amdgcn" in f]
    if not co:
        raise SystemExit("no device code object in " + obj)
    out = subprocess.check_output([OBJDUMP, "-d", os.path.join(d, co[0])], text=True)
    subprocess.call(["rm", "-rf", d])
    return out


def classify(op):
    if op.startswith("v_mad_u64_u32") or op.startswith("v_mad_i64_i32"):
        return "mad64"
    if op.startswith("v_accvgpr"):
        return "agpr"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_setpc") or op.startswith("s_swappc"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    return "other"


def main():
    obj, pat = sys.argv[1], sys.argv[2]
    text = disassemble(obj)
    funcs = re.split(r"\n(?=[0-9a-f]{16} <)", text)
    for f in funcs:
        head = f.split("\n", 1)[0]
        if pat not in head:
            continue
        name = head.split("<", 1)[1].rstrip(">:")
        ins = []   # (addr, op, operands)
        for line in f.split("\n")[1:]:
            m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)", line)
            if not m:
                continue
            ins.append((int(m.group(3), 16), m.group(1), m.group(2) + " " + m.group(4)))
        addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
        print(f"== {name}: {len(ins)} instructions")
        tot = collections.Counter(classify(op) for _, op, _ in ins)
        print("   whole kernel:", dict(tot))
        loops = []
        for i, (a, op, args) in enumerate(ins):
            if op.startswith("s_cbranch") or op == "s_branch":
                # objdump prints the target as a label-less offset: decode simm16 from the raw word is not shown here, so use the
                # symbolic "<name+0x...>" target it prints after the operand
                m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", args)
                if not m:
                    continue
                tgt_off = int(m.group(1), 16)
                base = ins[0][0]
                tgt = base + tgt_off
                if tgt <= a and tgt in addr_index:
                    loops.append((addr_index[tgt], i))
        loops.sort(key=lambda p: p[1] - p[0])
        for lo, hi in loops:
            c = collections.Counter(classify(op) for _, op, _ in ins[lo:hi + 1])
            n = hi - lo + 1
            valu = c["mad64"] + c["valu"] + c["agpr"]
            share = c["mad64"] / valu if valu else 0.0
            print(f"   loop [{lo:6d},{hi:6d}] {n:6d} instr  mad64 {c['mad64']:5d}  valu {c['valu']:5d}  agpr {c['agpr']:5d}  salu {c['salu']:4d}  "
                  f"lds {c['lds']:4d}  vmem {c['vmem']:4d}  wait {c['wait']:3d}  branch {c['branch']:3d}   mad share of VALU {share:.3f}")


if __name__ == "__main__":
    main()
