#!/bin/bash
# scripts/r6/isa_count.sh "<extra hipcc flags>" [kernel-substring] [source]: compile ba_linearize.hip for gfx950 with the library's flags + extras, print registers and the loop instruction mix
# (scripts/r6/isa_loop.py) of one kernel - the VALU count of the point loop is the figure of merit of the issue-bound linearisation
set -e
cd "$(dirname "$0")/../.."
EXTRA=$1; K=${2:-k_linearize_batchILi1E}; SRC=${3:-ldso_amd/csrc/ba_linearize.hip}
T=$(mktemp -d /tmp/isa.XXXX)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Wno-unused-value -Ildso_amd/csrc -Iinclude $EXTRA -c "$SRC" -o $T/k.o
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy --dump-section .hip_fatbin=$T/k.fat $T/k.o
$B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/k.fat --output=$T/k.co --unbundle
$B/llvm-objdump -d $T/k.co > $T/k.s
$B/llvm-readelf --notes $T/k.co | awk '/\.name:/ {name=$2} /\.sgpr_count:/ {s=$2} /\.vgpr_count:/ {v=$2} /\.vgpr_spill_count:/ {sp=$2} /\.wavefront_size:/ { if (name ~ /linearize_(batch|one)/) printf "%-60s vgpr %3d sgpr %3d spill %d\n", substr(name,1,60), v, s, sp }'
python scripts/r6/isa_loop.py $T/k.s "$K" ${MINB:-1000}
echo "disassembly: $T/k.s"
