#!/bin/bash
# VGPR / AGPR / SGPR / scratch / LDS / spills of every kernel in the objects of ldso_amd/libldso_hip.so (code-object metadata notes)
B=/opt/rocm/lib/llvm/bin
for o in /root/repo/ldso_amd/_obj/*.o; do
  $B/llvm-objcopy --dump-section .hip_fatbin=/tmp/_k.fat $o 2>/dev/null || continue
  $B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=/tmp/_k.fat --output=/tmp/_k.co --unbundle 2>/dev/null || continue
  $B/llvm-readelf --notes /tmp/_k.co 2>/dev/null | awk -v f=$(basename $o .hip.o) '
    /\.agpr_count:/ {a=$2} /\.group_segment_fixed_size:/ {l=$2} /\.name:/ {name=$2} /\.private_segment_fixed_size:/ {p=$2} /\.sgpr_count:/ {s=$2}
    /\.vgpr_count:/ {v=$2} /\.vgpr_spill_count:/ {sp=$2} /\.wavefront_size:/ { printf "%-14s %-90s vgpr %3d agpr %3d sgpr %3d scratch %5d lds %6d spill %d\n", f, substr(name,1,90), v, a, s, p, l, sp }'
done
