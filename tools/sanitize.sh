#!/bin/bash
# tools/sanitize.sh -- AddressSanitizer + UndefinedBehaviorSanitizer over everything that runs on the CPU (never on the GPU:
# the pool refuses GPU sanitizer runs): the oracle (oracle/pt_oracle.c), the C++ host mirror (host/*.hpp through pt_host_c.cpp), the
# device leaf headers compiled as host C++ (tests/hostshim), the C++ tile exchange and the host BVH builders (csrc/pt_lbvh.cpp).
# Works on a scratch copy of the repo so the normal build products stay untouched.   usage: bash tools/sanitize.sh [pytest args]
set -e -o pipefail
SRC=$(cd "$(dirname "$0")/.." && pwd)
DST=${PT_SANITIZE_DIR:-/tmp/pt_sanitize}
rm -rf "$DST"; mkdir -p "$DST"
tar -C "$SRC" --exclude=.git --exclude=gpurun_out --exclude='*.so' --exclude='*.o' --exclude=__pycache__ --exclude=.pytest_cache -cf - . | tar -C "$DST" -xf -
cp "$SRC"/directx-raytracing-spheres-demo_amd/libpt_hip.so "$DST"/directx-raytracing-spheres-demo_amd/ 2>/dev/null || true  # (loaded, never called, by the ABI tests)
cd "$DST"
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g"
export PT_EXTRA_CFLAGS="$SAN"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
# 1. the host BVH builders, standalone
g++ -std=c++20 -O1 -Wall $SAN -I include tests/cpp/lbvh_sanitize.cpp -o lbvh_sanitize && ./lbvh_sanitize
# 2. the C++ tile exchange
g++ -std=c++20 -O1 -Wall $SAN -I directx-raytracing-spheres-demo_amd/host tests/cpp/tile_exchange_test.cpp -o tile_exchange_test && ./tile_exchange_test
# 3. the CPU test-suite on sanitized libraries (the interpreter itself is not instrumented: the runtime is preloaded)
PRE=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
LD_PRELOAD=$PRE python3 -m pytest tests -m "not gpu" -x -q -k "not sanitizer and not gloo" "$@"
echo "sanitize: clean"
