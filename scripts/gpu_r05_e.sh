#!/bin/bash
# Round 5, GPU call E: GF cooperative multi-sweep launch (A/B at BASELINE configs[1] + its bit-identity test), R-MAT scale 22 SGNS at the planner's
# new width (751 wavefronts) next to 1536 on the same box.
mkdir -p gpurun_out/r05_rmat22
python scripts/ab_gf_fused.py > gpurun_out/r05_ab_gf_fused.jsonl 2> gpurun_out/r05_ab_gf_fused.err; cat gpurun_out/r05_ab_gf_fused.jsonl; tail -3 gpurun_out/r05_ab_gf_fused.err
timeout 600 python -m pytest tests/test_gf_gpu.py -m gpu -q -k "fused or rows_per_wave" 2>&1 | tail -5
python scripts/check_rmat17_launches.py --scale 22 --edges 64000000 --big 2048 --launches 0 --widths 0,1536,0,1536 --width-layouts 27 --width-launches 1 --out gpurun_out/r05_rmat22 --tag _e > gpurun_out/r05_rmat22_e.log 2>&1
tail -5 gpurun_out/r05_rmat22_e.log | cut -c1-200
