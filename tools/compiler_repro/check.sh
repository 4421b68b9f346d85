#!/bin/sh
# Builds the reproducers of tools/compiler_repro/README.md.  Static checks (ISA) run anywhere hipcc does; with a GPU the two
# run-time reproducers are executed too (each under `timeout`: the first one faults the queue when it reproduces).
cd "$(dirname "$0")"
B=${TMPDIR:-/tmp}/compiler_repro; mkdir -p $B
OBJDUMP=/opt/rocm/lib/llvm/bin/llvm-objdump
isa() { hipcc -O3 --offload-arch=gfx950 -std=c++17 -c $1 -o $B/x.o 2>/dev/null && (cd $B && rm -f x.o.* && $OBJDUMP --offloading x.o >/dev/null 2>&1; $OBJDUMP -d x.o.0.hipv4-amdgcn-amd-amdhsa--gfx950); }
echo "== mask_dropped_in_mad64 (static)"
isa mask_dropped_in_mad64.hip > $B/mask.s
if grep -q "v_bfe_u32\|0x7fffffff" $B/mask.s; then echo "   fixed: the mask is in the ISA"; else echo "   REPRODUCES: no mask ahead of the address multiply-add:"; grep -n "v_lshrrev_b32.* 7, \|v_mad_u64_u32" $B/mask.s | head -3 | sed 's/^/     /'; fi
echo "== sext_zext_mad (static): multiply-adds / moves per kernel (expected about 1458 / few)"
isa sext_zext_mad.hip > $B/sext.s
awk '/^[0-9a-f]+ <_Z1kILb/ {name=$2} /v_mad_u64_u32|v_mad_i64_i32/ {mad[name]++} /v_mov_b32/ {mov[name]++} END {for (n in mad) printf "   %s multiply-adds %d  v_mov %d\n", n, mad[n], mov[n]}' $B/sext.s
if command -v rocminfo >/dev/null 2>&1 && rocminfo 2>/dev/null | grep -q gfx950; then
  hipcc -O3 --offload-arch=gfx950 -std=c++17 mask_dropped_in_mad64.hip -o $B/mask 2>/dev/null && { echo "== mask_dropped_in_mad64 (run)"; timeout 60 $B/mask 2>&1 | tail -2; }
  hipcc -O3 --offload-arch=gfx950 -std=c++17 nested_switch_vm.hip -o $B/nested 2>/dev/null && { echo "== nested_switch_vm (run)"; timeout 120 $B/nested 2>&1 | tail -1; }
fi
