#!/bin/bash
mkdir -p gpurun_out/${OUT:-init_ab}
O=gpurun_out/${OUT:-init_ab}
timeout 600 python -m pytest tests/test_init_gpu.py -m gpu -q -x -p no:cacheprovider > $O/pytest_init.log 2>&1; echo "pytest_init rc=$?" > $O/rc.txt
for v in ${VARIANTS:-new}; do
  if [ $v = new ]; then unset LDSO_HIP_LIB; else export LDSO_HIP_LIB=$PWD/ldso_amd/libldso_hip_$v.so; fi
  timeout 300 python scripts/bench_init.py 640 480 6 > $O/bench_init_$v.log 2>&1; echo "$v rc=$?" >> $O/rc.txt
done
unset LDSO_HIP_LIB
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ini -o ini --output-format csv -- python $R/scripts/bench_init.py 640 480 6 > $R/$O/rocprof.log 2>&1; echo "rocprof rc=$?" >> $R/$O/rc.txt
cd $R
find /tmp/prof_ini -name "*kernel_stats.csv" -exec cp {} $O/init_kernel_stats.csv \;
cat $O/rc.txt; tail -3 $O/pytest_init.log; grep -h "^[0-9] snapped\|sweeps" $O/bench_init_*.log | cut -c1-260
