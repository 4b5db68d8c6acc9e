#!/bin/bash
# Round 6, GPU call T: is the -1.5 % of R-MAT scale 20 a CONCURRENCY effect at all?  The same pass at 64 and 128 wavefronts (and 768, 1536 with the final library).
O=gpurun_out/r06t
mkdir -p $O
timeout 2400 python scripts/sweep_width_schedule.py --scale 20 --flags 27 --repeats 1 --out $O/low20.jsonl --schedules '1:64;1:128;1:768;1:1536' > $O/low20.log 2>&1
cat $O/low20.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    if 'schedule' in r: print('scale %d %-8s: %+.2f %% (se %.2f)  sgns %.2f s %s' % (r['scale'], r['schedule'], r['gap_pct'], r['gap_se_pct'], r['sgns_s'], r['waves_and_hot_threshold']))
"
