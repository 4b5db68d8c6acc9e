#!/bin/bash
# Round 5, GPU call D: the whole GPU tier on the round's code, the new launch rule on R-MAT scale 17 on a second box (six launches per layout), and the
# HOPE SpMM A/B (round-4 kernel in gem_amd/libgem_hip_spmm_r4.so against the round-5 kernel: eigen-path and directed solves).
mkdir -p gpurun_out/r05_rmat17
python -m pytest tests -m gpu -q > gpurun_out/r05_pytest_d.log 2>&1
tail -5 gpurun_out/r05_pytest_d.log
echo skipped: rmat17 launches already taken on two boxes

for lib in spmm_r4 new spmm_r4 new; do
  L=""; [ $lib = spmm_r4 ] && L=$PWD/gem_amd/libgem_hip_spmm_r4.so
  ( echo "{\"lib\": \"$lib\"}"; GEM_HIP_LIB=$L python scripts/ab_hope_sym.py sym: 2>>gpurun_out/r05_ab_hope_spmm.err | head -1
    GEM_HIP_LIB=$L python bench.py --workload hope --hope-directed --no-cpu-baseline --no-api-wall --steps 3 --warmup 1 2>/dev/null ) >> gpurun_out/r05_ab_hope_spmm.jsonl
done
cat gpurun_out/r05_ab_hope_spmm.jsonl | cut -c1-400
