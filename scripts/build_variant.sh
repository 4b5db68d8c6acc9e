#!/bin/bash
# Builds an A/B / profiling variant of the library next to the product one:  scripts/build_variant.sh <suffix> <extra hipcc flags...>
# -> gem_amd/libgem_hip_<suffix>.so  (select with GEM_HIP_LIB=...).  Only the node2vec translation units (n2v.hip and the three SGNS
# instantiation files around sgns.hpp) are recompiled with the extra flags, e.g. -DGEMHIP_SGNS_PROFILE.
set -e
cd "$(dirname "$0")/.."
suf=$1; shift
mkdir -p gem_amd/build/$suf
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value -Wno-unused-result"
for f in n2v sgns_hogwild sgns_det sgns_part; do
    /opt/rocm/bin/hipcc $F "$@" -c gem_amd/csrc/$f.hip -o gem_amd/build/$suf/$f.hip.o &
done
wait
python -m gem_amd.build > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gem_amd/libgem_hip_$suf.so gem_amd/build/eval.hip.o gem_amd/build/gf.hip.o gem_amd/build/hope.hip.o gem_amd/build/runtime.hip.o \
    gem_amd/build/multi.hip.o gem_amd/build/$suf/n2v.hip.o gem_amd/build/$suf/sgns_hogwild.hip.o gem_amd/build/$suf/sgns_det.hip.o gem_amd/build/$suf/sgns_part.hip.o -ldl
echo gem_amd/libgem_hip_$suf.so
