#!/bin/bash
# round 4, GPU call K: the block-parallel second pass of the column arg-max, workspace zeroed on allocation only, and the host timeline of an eigen-path solve
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04k; mkdir -p $O
( timeout 300 python -m pytest tests/test_hope_kernels_gpu.py tests/test_hope_gpu.py tests/test_lap_gpu.py tests/test_run_karate_gpu.py -q -m gpu 2>&1 | tail -5 ) > $O/pytest_hope.log 2>&1; tail -2 $O/pytest_hope.log
( timeout 200 python scripts/ab_hope_sym.py timeline:GEMHIP_HOPE_DEBUG=1 plain: ) > $O/ab_hope_timeline.jsonl 2> $O/ab_hope_timeline.err
grep "host timeline" $O/ab_hope_timeline.err; cut -c1-250 $O/ab_hope_timeline.jsonl
for i in 1 2; do timeout 200 python bench.py --workload hope --steps 5 --warmup 1 --no-cpu-baseline --no-api-wall 2>> $O/bench_hope.log | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(json.dumps({'workload': d['config']['workload'], 'ms_per_step': d['ms_per_step'], 'device_window_s': d['roofline'].get('device_seconds_per_step'), 'spmm_s': d['roofline'].get('spmm_seconds_per_step')}))
"; done | tee $O/bench_hope.jsonl
