#!/bin/bash
# Round 6, GPU call E: with every Hogwild write lossless (call D) the gap got WORSE -- lost updates were masking over-application.  (1) who outranks the true
# neighbour of the "canary" nodes (oracle AP 1, AP here <= 0.5: mostly nodes of two-node components, count 800)?  (2) lossless writes + fresh reads.
set -x
O=gpurun_out/r06e
mkdir -p $O
timeout 900 python scripts/sweep_width_schedule.py --scale 17 --repeats 1 --out $O/canaries17.jsonl --canaries 12 --schedules '1:1536;1:64' > $O/canaries17.log 2>&1
for fr in 3 7; do
  timeout 900 python scripts/sweep_width_schedule.py --scale 17 --repeats 3 --fresh $fr --out $O/sched17_fresh$fr.jsonl --schedules '1:768;1:256' > $O/sched17_fresh$fr.log 2>&1
done
timeout 900 python scripts/sweep_width_schedule.py --scale 20 --repeats 2 --fresh 3 --out $O/sched20_fresh3.jsonl --schedules '1:768' > $O/sched20_fresh3.log 2>&1
cat $O/sched17_fresh*.jsonl $O/sched20_fresh3.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    if 'schedule' in r: print('scale %d fresh %d %-12s: %+.2f %% (se %.2f)  sgns %.2f s' % (r['scale'], r['fresh'], r['schedule'], r['gap_pct'], r['gap_se_pct'], r['sgns_s']))
"
grep canary $O/canaries17.jsonl | head -30
