#!/bin/bash
# Round 6, GPU call C: calls A and B refuted stale hub gradients and lost negative-row updates (profiles/r06_fresh_hot_rows_call_{a,b}.jsonl).  WHEN does the
# width cost quality -- in the first tokens (every wavefront computes its gradients from the initial tables at the highest alpha), at the end, or throughout?
# Width SCHEDULES over one pass, per-node APs kept for offline analysis.
set -x
O=gpurun_out/r06c
mkdir -p $O
timeout 1500 python scripts/sweep_width_schedule.py --scale 17 --repeats 3 --out $O/sched17.jsonl --save-ap $O/ap17 \
  --schedules '1:768;1:256;0.05:64,1:768;0.2:128,1:768;0.9:768,1:64;0.5:768,1:128;1:1536' > $O/sched17.log 2>&1
timeout 600 python scripts/sweep_width_schedule.py --scale 17 --repeats 2 --out $O/sched17_w64.jsonl --save-ap $O/ap17_w64 --schedules '1:64' > $O/sched17_w64.log 2>&1
timeout 1500 python scripts/sweep_width_schedule.py --scale 20 --repeats 2 --out $O/sched20.jsonl --save-ap $O/ap20 \
  --schedules '1:768;0.05:64,1:768;0.1:128,1:1536' > $O/sched20.log 2>&1
cat $O/sched17.jsonl $O/sched17_w64.jsonl $O/sched20.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('scale %d %-22s: %+.2f %% (se %.2f)  sgns %.2f s %s' % (r['scale'], r['schedule'], r['gap_pct'], r['gap_se_pct'], r['sgns_s'], r['waves_and_hot_threshold']))
"
