#!/bin/bash
# A/B library for the round-5 SpMM change: the current sources with hope.hip's hope_spmm16_kernel and its dispatch taken from round 4's closing commit
# (a5ddd28) -> gem_amd/libgem_hip_spmm_r4.so (select with GEM_HIP_LIB=...).
set -e
cd "$(dirname "$0")/.."
mkdir -p gem_amd/build/spmm_r4
python - <<'PY'
import subprocess
cur = open('gem_amd/csrc/hope.hip').read()
old = subprocess.check_output(['git', 'show', 'a5ddd28:gem_amd/csrc/hope.hip']).decode()
def cut(s, a, b):
    i = s.index(a); j = s.index(b, i)
    return i, j
A1, B1 = "template <int CPL16, int U>\n__global__ __launch_bounds__(256) void hope_spmm16_kernel(", "// ------------------------------------------------------- Gram  P[slab] = X^T Y  (MFMA fp32)"
i, j = cut(cur, A1, B1); oi, oj = cut(old, A1, B1)
cur = cur[:i] + old[oi:oj] + cur[j:]
A2, B2 = "#define SPMM16(C, U) hipLaunchKernelGGL(", "#undef SPMM16_BY_U"
i, j = cut(cur, A2, B2); oi, oj = cut(old, A2, B2)
cur = cur[:i] + old[oi:oj] + cur[j:]
open('gem_amd/build/spmm_r4/hope.hip', 'w').write(cur)
PY
cp gem_amd/csrc/common.hpp gem_amd/build/spmm_r4/
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value -Wno-unused-result -Iinclude -Igem_amd/csrc -c gem_amd/build/spmm_r4/hope.hip -o gem_amd/build/spmm_r4/hope.hip.o
python -m gem_amd.build > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gem_amd/libgem_hip_spmm_r4.so gem_amd/build/eval.hip.o gem_amd/build/gf.hip.o gem_amd/build/spmm_r4/hope.hip.o gem_amd/build/runtime.hip.o \
    gem_amd/build/multi.hip.o gem_amd/build/n2v.hip.o gem_amd/build/sgns_hogwild.hip.o gem_amd/build/sgns_det.hip.o gem_amd/build/sgns_part.hip.o -ldl
echo gem_amd/libgem_hip_spmm_r4.so
