#!/bin/bash
# A/B of k_linearize builds on one box: scripts/exp_lin_variants.sh old base pf ...   (ldso_amd/libldso_hip_<name>.so; base = the product library)
for L in "$@"; do
  if [ "$L" = "base" ]; then unset LDSO_HIP_LIB; else export LDSO_HIP_LIB=$PWD/ldso_amd/libldso_hip_$L.so; fi
  echo "=== $L"
  bash scripts/exp_lin.sh 2>&1 | grep -v "desc=1"
done
