#!/bin/bash
# Round 6, GPU call W: R-MAT scale 22 with the FINAL library at the planner's width (both layouts) and at 768, per-node APs kept for the pairing with the oracle goldens.
O=gpurun_out/r06w
mkdir -p $O
timeout 1500 python scripts/sweep_width_schedule.py --scale 22 --flags 27 --repeats 1 --out $O/sched22_f27.jsonl --save-ap $O/ap22 --schedules '1:0;1:768' > $O/sched22_f27.log 2>&1
timeout 900 python scripts/sweep_width_schedule.py --scale 22 --flags 11 --repeats 1 --out $O/sched22_f11.jsonl --save-ap $O/ap22 --schedules '1:0' > $O/sched22_f11.log 2>&1
cat $O/sched22_f27.jsonl $O/sched22_f11.jsonl | cut -c1-260
