#!/bin/bash
# The CPU test suite against a SANITIZER build of the library's host side (AddressSanitizer + UBSan; device code untouched — GPU ASan
# is not available on this pool): the artifact readers (JSON / CBOR), the bytecode compilers, the code generator of the specialised
# kernels, the host verifier and the placement code run under it. usage: tools/asan_cpu_suite.sh [pytest args]
set -e
cd "$(dirname "$0")/.."
if [ "$1" = oracle ]; then
  # the CHECKER under gcc's sanitizers (oracle/_build_san/), the product library as built: the oracle-only and oracle-vs-host tests
  shift
  export ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:halt_on_error=1
  export UBSAN_OPTIONS=print_stacktrace=0:halt_on_error=0
  LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" POWDR_ORACLE_SAN=1 python -m pytest tests -m "not gpu" -x -q -s "$@"
  exit $?
fi
POWDR_BUILD_ASAN=1 python -m powdr_amd.build > /dev/null
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
export ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:abort_on_error=0:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=0:halt_on_error=0   # every finding of the run is reported (stderr); ASan findings stop it
LD_PRELOAD=$RT POWDR_LIB_DIR=$PWD/powdr_amd/lib_asan python -m pytest tests -m "not gpu" -x -q -s "$@"
