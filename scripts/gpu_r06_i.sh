#!/bin/bash
# Round 6, GPU call I: the test subset of call H again (RCCL's 1-rank init failed there with "unhandled cuda error" after 75 other tests of the process; a sticky
# HIP error is now cleared before ncclCommInitAll), without -x, RCCL warnings on.
set -x
O=gpurun_out/r06i
mkdir -p $O
NCCL_DEBUG=WARN timeout 2400 python -m pytest tests/test_n2v_gpu.py tests/test_rmat_gpu.py tests/test_multi_capi_gpu.py tests/test_run_sbm_gpu.py -m gpu -q > $O/pytest_subset.log 2>&1
tail -60 $O/pytest_subset.log | cut -c1-400
