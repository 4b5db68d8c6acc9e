#!/bin/bash
# Round 6, GPU call V: the candidate rule (concurrent-touch bound 0.10, floor 128 wavefronts): R-MAT scale 20 at 126 wavefronts and scale 17 at 128, both layouts.
O=gpurun_out/r06v
mkdir -p $O
timeout 1500 python scripts/sweep_width_schedule.py --scale 20 --flags 27 --repeats 3 --out $O/cand20.jsonl --schedules '1:126' > $O/cand20_f27.log 2>&1
timeout 1500 python scripts/sweep_width_schedule.py --scale 20 --flags 11 --repeats 2 --out $O/cand20.jsonl --schedules '1:126' > $O/cand20_f11.log 2>&1
for fl in 27 11; do timeout 600 python scripts/sweep_width_schedule.py --scale 17 --flags $fl --repeats 2 --out $O/cand17.jsonl --schedules '1:128' > $O/cand17_f$fl.log 2>&1; done
cat $O/cand20.jsonl $O/cand17.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    if 'schedule' in r: print('scale %d flags %d %-6s: %+.2f %% (se %.2f)  sgns %.2f s' % (r['scale'], r['flags'], r['schedule'], r['gap_pct'], r['gap_se_pct'], r['sgns_s']))
"
