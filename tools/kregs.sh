#!/bin/sh
# registers / scratch of the kernels in one object: tools/kregs.sh build/msm_inst_mnt4g1.o [filter]
O=$1; F=${2:-.}
D=$(mktemp -d); cp $O $D/x.o; cd $D
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o >/dev/null 2>&1
/opt/rocm/lib/llvm/bin/llvm-readelf --notes x.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 | grep -E "^\s+\.name:|\.vgpr_count|\.private_segment_fixed_size|\.agpr_count|vgpr_spill" | sed 's/^ *//' | paste - - - - - | sed 's/_ZN6mnt753//' | grep "$F" | cut -c1-220
rm -rf $D
