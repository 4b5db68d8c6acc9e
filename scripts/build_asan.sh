#!/bin/bash
# AddressSanitizer build of the HOST side of libgem_hip.so (the C-ABI shim, the launch planner, the host eigensolvers, the Vose / CSR / level-schedule
# builders) -- device code is compiled as usual (-fno-gpu-sanitize).  SURVEY section 5 asked for a sanitizer build of the host shim:
#
#   scripts/build_asan.sh            # -> gem_amd/libgem_hip_asan.so
#   scripts/build_asan.sh test       # ... and runs the host-only tests against it (no GPU needed): tests/test_sgns_plan.py, tests/test_capi.py
#
# With a GPU the whole -m gpu tier runs against it the same way (LD_PRELOAD + GEM_HIP_LIB as below); expect it to be several times slower.
set -e
cd "$(dirname "$0")/.."
mkdir -p gem_amd/build/asan
for f in eval gf hope runtime n2v sgns_hogwild sgns_det sgns_part multi; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer -w \
        -c gem_amd/csrc/$f.hip -o gem_amd/build/asan/$f.hip.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -o gem_amd/libgem_hip_asan.so gem_amd/build/asan/*.o -ldl
echo gem_amd/libgem_hip_asan.so
if [ "$1" = test ]; then
    ASAN=$(find /opt/rocm/lib/llvm -name "libclang_rt.asan-x86_64.so" | head -1)
    LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0 GEM_HIP_LIB=$PWD/gem_amd/libgem_hip_asan.so python -m pytest tests/test_sgns_plan.py tests/test_capi.py -q
fi
