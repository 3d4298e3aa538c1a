#!/bin/bash
# AddressSanitizer pass over the host-side C++ the tests load: the reference-compiled pin (oracle/ref_driver.cc + the reference's translation units) and the
# compiled adapter (adapter/*.cc).  Build first, where /root/reference exists:   make -C oracle ref_asan && make -C adapter asan
# then (CPU: the pin tests; GPU box: add the adapter tests)
#   scripts/asan_check.sh                      # tests/test_ref_pin.py tests/test_adapter_sequence_cpu.py, no GPU
#   gpurun -- 'scripts/asan_check.sh gpu'      # + tests/test_activate_gpu.py tests/test_adapter_gpu.py tests/test_adapter_sequence_gpu.py tests/test_adapter_threads_gpu.py tests/test_trace_gpu.py
# gcc's libasan has to be the first library of the process (python itself is not instrumented): LD_PRELOAD.  pytest -s: the report goes to stderr of the process.
# libldso_hip.so is NOT instrumented (a clang host-ASan build of it runs out of memory inside an HSA interceptor on this image).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export LDSO_ADAPTER_LIB=$PWD/adapter/_build_asan/libldso_adapter_test.so
export LDSO_REF_LIB=$PWD/oracle/_ref/asan/libldso_ref.so
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=1:detect_odr_violation=0:alloc_dealloc_mismatch=0
RT=$(gcc -print-file-name=libasan.so)
if [ "${1:-cpu}" = gpu ]; then
    T="tests/test_activate_gpu.py tests/test_adapter_gpu.py tests/test_adapter_sequence_gpu.py tests/test_adapter_threads_gpu.py tests/test_trace_gpu.py"; M="gpu"
else
    T="tests/test_ref_pin.py tests/test_adapter_sequence_cpu.py"; M="not gpu"
fi
LD_PRELOAD=$RT timeout 1500 python -m pytest $T -m "$M" -q -x -s -p no:cacheprovider > gpurun_out/asan_check_${1:-cpu}.log 2>&1
rc=$?
grep -n "ERROR: AddressSanitizer\|SUMMARY: AddressSanitizer" gpurun_out/asan_check_${1:-cpu}.log
tail -3 gpurun_out/asan_check_${1:-cpu}.log
echo "asan_check rc=$rc"
exit $rc
