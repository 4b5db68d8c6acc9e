#!/bin/bash
# Round 6, GPU call H: the planner with the floor of 256 wavefronts and locally hot rows -- the node2vec / R-MAT / multi-GPU-plugin / HOPE-verbose tests, and
# R-MAT scale 22 passes at the planner's width and around it with the per-node APs kept (the scale-22 oracle runs finish later tonight: paired offline).
set -x
O=gpurun_out/r06h
mkdir -p $O
timeout 1500 python -m pytest tests/test_n2v_gpu.py tests/test_rmat_gpu.py tests/test_multi_capi_gpu.py tests/test_run_sbm_gpu.py -m gpu -x -q > $O/pytest_subset.log 2>&1
tail -25 $O/pytest_subset.log
timeout 1500 python scripts/sweep_width_schedule.py --scale 22 --flags 27 --repeats 1 --out $O/sched22_f27.jsonl --save-ap $O/ap22 --schedules '1:0;1:256;1:768' > $O/sched22_f27.log 2>&1
timeout 900 python scripts/sweep_width_schedule.py --scale 22 --flags 11 --repeats 1 --out $O/sched22_f11.jsonl --save-ap $O/ap22 --schedules '1:0' > $O/sched22_f11.log 2>&1
cat $O/sched22_f27.jsonl $O/sched22_f11.jsonl
tail -3 $O/sched22_f27.log
