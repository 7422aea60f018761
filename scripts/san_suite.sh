#!/bin/bash
# The CPU suite against the sanitizer build of the library's host side (ASan + UBSan, `make SAN=1`): the counterpart of the reference's -DDEV=1 build
# (CMakeLists.txt:20-24).  CPU only: GPU ASan / xnack+ is not available on this pool.  usage: scripts/san_suite.sh [pytest args]   -> profiles/san_suite_latest.txt
set -u
cd "$(dirname "$0")/.."
make -s -j8 -C centrolign_amd/csrc SAN=1 2>&1 | grep -v 'loop not unrolled' | grep -E 'error|Error' && exit 1
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
export CL_LIBRARY=$PWD/centrolign_amd/lib/san/libcentrolign_amd.so
# python itself is not instrumented: no leak report at exit (the interpreter never frees its arenas), no ODR check across the HIP runtime's copies of libstdc++
export ASAN_OPTIONS=detect_leaks=0:detect_odr_violation=0:abort_on_error=0:halt_on_error=1:allocator_may_return_null=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
LD_PRELOAD=$RT python -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@" 2>&1 | tee profiles/san_suite_latest.txt | tail -15
