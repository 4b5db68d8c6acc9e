#!/bin/bash
# round 4, GPU call B: partitioned kernel with whole-walk mode + duty-scaled width, C-ABI multi-GPU entry points, gf.cpp binary pin
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04b; mkdir -p $O
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_n2v_partitioned_gpu.py tests/test_multi_capi_gpu.py -q -m gpu 2>&1 | tail -40 ) > $O/pytest_part_multi.log 2>&1
( timeout 600 python -m pytest tests/test_gf_gpu.py -q -m gpu -k "rows_per_wave or gf_cpp_binary" 2>&1 | tail -8 ) > $O/pytest_gf.log 2>&1
( timeout 1200 python scripts/check_partitioned_1m.py 2 4 8 ) > $O/partitioned_1m.jsonl 2> $O/partitioned_1m.err
tail -5 $O/pytest_part_multi.log $O/pytest_gf.log; cut -c1-700 $O/partitioned_1m.jsonl; tail -2 $O/partitioned_1m.err
