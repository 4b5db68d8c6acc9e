#!/bin/bash
# round 4, GPU call D: fixed tests, bucket kernel visiting only centre positions, GF striped rows A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04d; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_lap_gpu.py tests/test_rmat_gpu.py tests/test_n2v_partitioned_gpu.py tests/test_multi_capi_gpu.py -q -m gpu 2>&1 | tail -60 ) > $O/pytest_fixed.log 2>&1
( timeout 300 python -m pytest tests/test_gf_gpu.py -q -m gpu -k "rows_per_wave" 2>&1 | tail -5 ) > $O/pytest_gf.log 2>&1
( timeout 300 python scripts/ab_gf_rows.py 8 8:1024 8:512 8:2048 4:1024 16:1024 8 ) > $O/ab_gf_stripe.jsonl 2> $O/ab_gf_stripe.err
( timeout 900 python scripts/check_partitioned_1m.py 4 8 ) > $O/partitioned_1m.jsonl 2> $O/partitioned_1m.err
tail -8 $O/pytest_fixed.log; tail -3 $O/pytest_gf.log; cat $O/ab_gf_stripe.jsonl; cut -c1-600 $O/partitioned_1m.jsonl
