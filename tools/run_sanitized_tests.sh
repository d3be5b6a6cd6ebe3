#!/bin/bash
# CPU-side sanitizers (GPU sanitizers are not available on this pool): builds the oracle and
# the host build of the kernel bodies with AddressSanitizer + UndefinedBehaviorSanitizer and runs the whole
# `not gpu` suite on them.  Output -> profiles/<tag>_asan_ubsan_cpu_suite.txt
#   usage: bash tools/run_sanitized_tests.sh [tag]
TAG=${1:-r04}
cd "$(dirname "$0")/.."
OUT=profiles/${TAG}_asan_ubsan_cpu_suite.txt
ASAN_LIB=$(gcc -print-file-name=libasan.so)
{
  echo "# $(date -u +%FT%TZ)  gcc $(gcc -dumpversion)  LD_PRELOAD=$ASAN_LIB  NMPC_SANITIZE=1"
  echo "# flags: -fsanitize=address,undefined -fno-sanitize-recover=undefined (oracle/Makefile asan, tests/hostsim)"
  # detect_leaks=0: CPython itself leaks at exit; every allocation of the C code is still checked for overflow / use-after-free
  LD_PRELOAD=$ASAN_LIB ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    NMPC_SANITIZE=1 OMP_NUM_THREADS=4 python -m pytest tests -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15
  echo "# loaded sanitizer builds:"; ls -la oracle/libnmpc_oracle_asan.so tests/hostsim/libnmpc_hostsim_asan.so
} > "$OUT" 2>&1
tail -8 "$OUT"
