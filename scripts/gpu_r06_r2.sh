#!/bin/bash
# Round 6, GPU call R2: the whole GPU suite again after the fix of the set-up race in multi.hip (Fabric::sync_all waits for the device, not only the rank's stream).
O=gpurun_out/r06r2
mkdir -p $O
timeout 3000 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu_full.log 2>&1; tail -30 $O/pytest_gpu_full.log | cut -c1-300
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_multi_capi_gpu.py -q -m gpu -k "gf_train_multi or rccl_selftest" > $O/multi_rep$i.log 2>&1; tail -1 $O/multi_rep$i.log; done
