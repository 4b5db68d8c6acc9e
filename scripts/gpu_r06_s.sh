#!/bin/bash
# Round 6, GPU call S: the shipped configuration (planner's width, final library) on R-MAT scale 17 and 20, four launches per layout, for the record.
O=gpurun_out/r06s
mkdir -p $O
for fl in 27 11; do
  timeout 900 python scripts/sweep_width_schedule.py --scale 17 --flags $fl --repeats 4 --out $O/final17.jsonl --schedules '1:0' > $O/final17_f$fl.log 2>&1
  timeout 1500 python scripts/sweep_width_schedule.py --scale 20 --flags $fl --repeats 4 --out $O/final20.jsonl --schedules '1:0' > $O/final20_f$fl.log 2>&1
done
cat $O/final17.jsonl $O/final20.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    if 'schedule' in r: print('scale %d flags %d: %+.2f %% (se %.2f)  sgns %.2f s %s' % (r['scale'], r['flags'], r['gap_pct'], r['gap_se_pct'], r['sgns_s'], r['waves_and_hot_threshold']))
"
