#!/bin/bash
# the tests that need no GPU against the sanitizer build of the host side (scripts/build_asan.sh)
cd "$(dirname "$0")/.."
LIB=$PWD/sequali_amd/build/asan/libsqgpu_asan.so
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
# python itself is not instrumented: leaks of the interpreter are not ours to report; an ODR check across the
# un-instrumented libamdhip64 is off
export ASAN_OPTIONS=detect_leaks=0:detect_odr_violation=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
SQ_LIB=$LIB LD_PRELOAD=$RT python -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@"
