#!/bin/bash
# Round 6, GPU call F: call E found the heavy tail -- ONE isolated edge whose two rows were trained by two wavefronts at once (norm 7.97 against 7.0 for its
# peers) outranks the true neighbour of dozens of collinear peers.  LOCALLY HOT ROWS (n2v.hip ensure_hotkey: nodes with >= 8 tokens per walk that contains
# them never enter the LDS window) against the sequential oracle.
set -x
O=gpurun_out/r06f
mkdir -p $O
timeout 1500 python scripts/sweep_width_schedule.py --scale 17 --repeats 4 --out $O/sched17.jsonl --save-ap $O/ap17 --canaries 4 --schedules '1:768;1:1536;1:256' > $O/sched17.log 2>&1
timeout 1500 python scripts/sweep_width_schedule.py --scale 20 --repeats 2 --out $O/sched20.jsonl --save-ap $O/ap20 --schedules '1:768;1:1536;1:256' > $O/sched20.log 2>&1
timeout 900 python scripts/sweep_width_schedule.py --scale 20 --flags 11 --repeats 2 --out $O/sched20_f11.jsonl --schedules '1:768' > $O/sched20_f11.log 2>&1
cat $O/sched17.jsonl $O/sched20.jsonl $O/sched20_f11.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    if 'schedule' in r: print('scale %d flags %d %-8s: %+.2f %% (se %.2f)  sgns %.2f s %s' % (r['scale'], r['flags'], r['schedule'], r['gap_pct'], r['gap_se_pct'], r['sgns_s'], r['waves_and_hot_threshold']))
"
grep -c canary $O/sched17.jsonl
