#!/bin/bash
# round 4, call N: the host eigensolver with two-column fused loops and the AVX-512 build chosen at run time -- HOPE / LE / LLE tests, then the directed solve
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r04n; mkdir -p $O
( timeout 120 python -m pytest tests/test_hope_gpu.py tests/test_lap_gpu.py -q -m gpu 2>&1 | tail -3 ) > $O/pytest_hope.log 2>&1; tail -1 $O/pytest_hope.log
for cfg in "avx2 0" "auto 1"; do set -- $cfg
  GEMHIP_EIG_ISA=$1 GEMHIP_EIG_PAIRS=$2 timeout 100 python bench.py --workload hope --hope-directed --steps 5 --warmup 1 --no-cpu-baseline --no-api-wall 2>> $O/bench.log | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(json.dumps({'eig_isa': '$1', 'eig_pairs': $2, 'workload': d['config']['workload'], 'ms_per_step': d['ms_per_step'], 'host_eig_s': d['roofline'].get('host_eig_seconds_per_step')}))"
done | tee $O/ab_directed_eig.jsonl
