#!/bin/bash
# round 4, GPU call E: bucket kernels with the as-loaded copies in global scratch (12 wavefronts per CU), LE test
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04e; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_lap_gpu.py tests/test_n2v_partitioned_gpu.py tests/test_multi_capi_gpu.py -q -m gpu 2>&1 | tail -40 ) > $O/pytest.log 2>&1
( timeout 900 python scripts/check_partitioned_1m.py 2 4 8 ) > $O/partitioned_1m.jsonl 2> $O/partitioned_1m.err
tail -6 $O/pytest.log; cut -c1-600 $O/partitioned_1m.jsonl
