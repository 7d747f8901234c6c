#!/bin/bash
# the parser's tests that need no GPU against the ThreadSanitizer build of the host side (scripts/build_tsan.sh)
cd "$(dirname "$0")/.."
LIB=$PWD/sequali_amd/build/tsan/libsqgpu_tsan.so
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.tsan-x86_64.so)
export TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0:exitcode=66:second_deadlock_stack=1
SQ_LIB=$LIB LD_PRELOAD=$RT python -m pytest -q -m "not gpu" -p no:cacheprovider "${@:-tests/test_parser_source_cpu.py}"
