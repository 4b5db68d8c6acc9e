#!/bin/bash
# round 4, call M (host code only; run on the GPU box because its host CPU -- AMD EPYC 9575F, Zen 5 -- is the one the solver's projected eigenproblems run on):
# the eigensolver's inner loops built for AVX2 / AVX-512, and the deferred-update (blocked) Householder reduction against the plain one
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r04m; mkdir -p $O
lscpu | grep -E "Model name|^CPU\(s\)|Thread|L2|L3" | cut -c1-200 > $O/lscpu.txt
for cfg in "avx2 0" "avx2 1" "avx512 0" "avx512 1"; do set -- $cfg; echo "ISA=$1 BLOCKED=$2"; GEMHIP_EIG_ISA=$1 GEMHIP_EIG_BLOCKED=$2 python - <<'PY'
import sys; sys.path.insert(0,'scripts'); sys.argv=['x']
import bench_host_eig as b
b.run(448,72,1); b.run(512,80,1); b.run(256,72,1)
PY
done > $O/host_eig_variants.txt 2>&1
cat $O/lscpu.txt $O/host_eig_variants.txt
