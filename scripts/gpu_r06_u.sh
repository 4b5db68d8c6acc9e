#!/bin/bash
# Round 6, GPU call U: R-MAT scale 22 at lower widths (332 = the concurrent-touch bound at 0.10; 128), per-node APs kept for pairing with the oracle runs that finish tonight.
O=gpurun_out/r06u
mkdir -p $O
timeout 1500 python scripts/sweep_width_schedule.py --scale 22 --flags 27 --repeats 1 --out $O/sched22_f27.jsonl --save-ap $O/ap22 --schedules '1:332;1:128' > $O/sched22_f27.log 2>&1
timeout 900 python scripts/sweep_width_schedule.py --scale 22 --flags 11 --repeats 1 --out $O/sched22_f11.jsonl --save-ap $O/ap22 --schedules '1:332' > $O/sched22_f11.log 2>&1
cat $O/sched22_f27.jsonl $O/sched22_f11.jsonl | cut -c1-260
