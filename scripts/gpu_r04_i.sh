#!/bin/bash
# round 4, GPU call I: LDS-staged tall-skinny GEMM / Ritz kernels (bit-identical to the one-wavefront-per-tile kernels): tests, then the A/B on both HOPE solves
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04i; mkdir -p $O
( timeout 600 python -m pytest tests/test_hope_kernels_gpu.py tests/test_hope_gpu.py tests/test_lap_gpu.py -q -m gpu 2>&1 | tail -15 ) > $O/pytest_hope.log 2>&1
tail -3 $O/pytest_hope.log
for v in 0 1 0 1; do
  for dirflag in "" "--hope-directed"; do
    GEMHIP_HOPE_TSGEMM_LDS=$v timeout 300 python bench.py --workload hope $dirflag --steps 5 --warmup 1 --no-cpu-baseline --no-api-wall 2>> $O/bench_hope_lds.log | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(json.dumps({'tsgemm_lds': $v, 'workload': d['config']['workload'], 'ms_per_step': d['ms_per_step'], 'value': d['value'], 'spmm_s': d['roofline'].get('spmm_seconds_per_step'), 'eig_s': d['roofline'].get('host_eig_seconds_per_step')}))
" >> $O/ab_hope_tsgemm_lds.jsonl
  done
done
cat $O/ab_hope_tsgemm_lds.jsonl
