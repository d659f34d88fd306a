#!/bin/bash
# The CPU-side test infrastructure (C oracle, host twin of the device math) under AddressSanitizer + UBSan: rebuilds both
# with -fsanitize=address,undefined, runs the CPU tests that drive them with the sanitizer runtime preloaded, then restores
# the ordinary builds.  (GPU AddressSanitizer is not available on the pool; the device code's fp32 math is what the twin compiles.)
set -e
cd "$(dirname "$0")/.."
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1"
mkdir -p oracle/_build tests/host_twin/_build
gcc $SAN -fPIC -std=c11 -ffp-contract=off -fno-fast-math -fopenmp -shared -o oracle/_build/libspacegym_oracle.so oracle/spacegym_oracle.c -lm
g++ $SAN -std=c++17 -fPIC -shared -march=haswell -ffp-contract=off -I space_gym_amd/csrc -o tests/host_twin/_build/libsg_host_twin.so tests/host_twin/twin.cpp
touch oracle/_build/libspacegym_oracle.so tests/host_twin/_build/libsg_host_twin.so
rc=0
LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  python -m pytest tests/test_oracle_golden.py tests/test_host_twin.py -x -q -p no:cacheprovider || rc=$?
# back to the ordinary builds
touch oracle/spacegym_oracle.c tests/host_twin/twin.cpp
python -c "import oracle, sys; oracle.build(); sys.path.insert(0, 'tests/host_twin'); import build; build.build()"
exit $rc
