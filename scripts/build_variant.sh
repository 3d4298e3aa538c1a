#!/bin/bash
# Build a variant of libldso_hip.so that differs in ONE translation unit compiled with extra flags (kernel experiments):
#   scripts/build_variant.sh <name> <file.hip> "<extra flags>" [<source path>]  ->  ldso_amd/libldso_hip_<name>.so   (select it with LDSO_HIP_LIB=...)
# <file.hip> names the translation unit of ldso_amd/csrc that is replaced; with <source path> the replacement is compiled from that file instead (e.g. the
# unit of another branch: git show next/x:ldso_amd/csrc/tracker.hip > /tmp/t.hip), with ldso_amd/csrc on the include path.
set -e
cd "$(dirname "$0")/.."
python -m ldso_amd.build > /dev/null
NAME=$1; SRC=$2; EXTRA=$3; FROM=${4:-ldso_amd/csrc/$SRC}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Wno-unused-value -Ildso_amd/csrc -Iinclude $EXTRA -c "$FROM" -o /tmp/variant_$NAME.o
OBJS=$(ls ldso_amd/_obj/*.o | grep -v "/$SRC.o")
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/variant_$NAME.o -o ldso_amd/libldso_hip_$NAME.so
echo ldso_amd/libldso_hip_$NAME.so
