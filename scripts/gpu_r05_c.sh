#!/bin/bash
# Round 5, GPU call C: the new launch rule (concurrent-touch bound) on R-MAT scale 17 -- six one-shot launches per layout -- and the test files this
# round touched.
mkdir -p gpurun_out/r05_rmat17
python scripts/check_rmat17_launches.py --launches 6 --widths '' --out gpurun_out/r05_rmat17 --tag _c > gpurun_out/r05_rmat17_c.log 2>&1
tail -13 gpurun_out/r05_rmat17_c.log
python -m pytest tests/test_rmat_gpu.py tests/test_multi_capi_gpu.py tests/test_run_sbm_gpu.py tests/test_gf_gpu.py tests/test_n2v_partitioned_gpu.py tests/test_bench_gpu.py -m gpu -q -x \
    -k "not headline_size and not at_100k" > gpurun_out/r05_pytest_c.log 2>&1
tail -30 gpurun_out/r05_pytest_c.log
