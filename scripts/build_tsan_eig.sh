#!/bin/bash
# ThreadSanitizer run of the threaded host eigensolver (hope.hip: eig_reduce_mt's spin barriers, the chunked back-transformation).  The host half of
# hope.hip is compiled with -fsanitize=thread (device code as usual), linked with a small driver (scripts/tsan/eig_driver.cpp: dense, diagonal --
# every step takes the zero-reflector branch -- and half-zero matrices at 2, 3 and 4 threads, partial and full solver) and run WITHOUT a GPU:
#
#   scripts/build_tsan_eig.sh        # prints the driver's lines and the number of ThreadSanitizer reports (expected: 0)
#
# (TSan slows the threads enough that the contended-host bail-out of eig_reduce_mt is taken too, so that path is covered as well.)
set -e
cd "$(dirname "$0")/.."
OUT=gem_amd/build/tsan
mkdir -p $OUT
[ -f gem_amd/build/runtime.hip.o ] || python -m gem_amd.build
CL=/opt/rocm/lib/llvm/bin/clang++
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=thread -w -c gem_amd/csrc/hope.hip -o $OUT/hope.o
$CL -fsanitize=thread -g -O1 -c scripts/tsan/eig_driver.cpp -o $OUT/driver.o
$CL -fsanitize=thread $OUT/driver.o $OUT/hope.o gem_amd/build/runtime.hip.o -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -o $OUT/eig_tsan
TSAN_OPTIONS="halt_on_error=0" timeout 900 $OUT/eig_tsan > $OUT/out.txt 2>&1 || { tail -30 $OUT/out.txt; echo "driver failed"; exit 1; }
tail -3 $OUT/out.txt
echo "ThreadSanitizer reports: $(grep -c 'WARNING: ThreadSanitizer' $OUT/out.txt || true)"
