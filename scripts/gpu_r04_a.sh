#!/bin/bash
# round 4, GPU call A: new kernels' parity tests + first measurements (GF rows-per-wave A/B, partitioned cost model, HOPE host-eig threads)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04a; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_n2v_partitioned_gpu.py tests/test_gf_gpu.py -x -q -m gpu 2>&1 | tail -15 ) > $O/pytest_new.log 2>&1
( timeout 600 python -m pytest tests/test_n2v_gpu.py -x -q -m gpu -k "deterministic or window_cache_equals" 2>&1 | tail -8 ) > $O/pytest_n2v.log 2>&1
( timeout 300 python scripts/ab_gf_rows.py 1 2 4 8 16 32 ) > $O/ab_gf_rows.jsonl 2> $O/ab_gf_rows.err
( timeout 900 python scripts/check_partitioned_1m.py 1 4 8 ) > $O/partitioned_1m.jsonl 2> $O/partitioned_1m.err
for t in 1 4; do
  ( GEMHIP_EIG_THREADS=$t timeout 300 python bench.py --workload hope --hope-directed --no-cpu-baseline --no-api-wall --steps 5 --warmup 1 ) > $O/hope_directed_eig$t.json 2> $O/hope_directed_eig$t.err
done
( timeout 600 python -m pytest tests/test_bench_gpu.py -x -q -m gpu -k "two_rank" 2>&1 | tail -8 ) > $O/pytest_bench2.log 2>&1
tail -3 $O/pytest_new.log $O/pytest_n2v.log $O/pytest_bench2.log; cat $O/ab_gf_rows.jsonl; cut -c1-400 $O/partitioned_1m.jsonl; tail -2 $O/partitioned_1m.err
