#!/bin/bash
# full-library variant: every translation unit with extra flags -> ldso_amd/libldso_hip_<name>.so
set -e
cd /root/repo
NAME=$1; EXTRA=$2
mkdir -p /tmp/objs_$NAME
for f in ldso_amd/csrc/*.hip; do
  b=$(basename $f)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Wno-unused-value -Ildso_amd/csrc -Iinclude $EXTRA -c $f -o /tmp/objs_$NAME/$b.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC /tmp/objs_$NAME/*.o -o ldso_amd/libldso_hip_$NAME.so
echo ldso_amd/libldso_hip_$NAME.so
