#!/bin/bash
# Round 6, GPU call D: the slow path of the pair step (a repeated target: nearly always a hub row) STORED its five rows -- a plain store of `row as re-fetched +
# my update` that wipes out every atomic add other wavefronts landed in between (~1 us).  Estimated on R-MAT scale 17: 5.4 % of the top hub's negative
# updates take that path and each destroys ~4 foreign updates at 768 wavefronts.  Now atomic adds (sgns.hpp); widths re-measured against the oracle.
set -x
O=gpurun_out/r06d
mkdir -p $O
timeout 1500 python scripts/sweep_width_schedule.py --scale 17 --repeats 3 --out $O/sched17.jsonl --save-ap $O/ap17 --schedules '1:768;1:1536;1:256' > $O/sched17.log 2>&1
timeout 1500 python scripts/sweep_width_schedule.py --scale 20 --repeats 2 --out $O/sched20.jsonl --save-ap $O/ap20 --schedules '1:768;1:1536' > $O/sched20.log 2>&1
timeout 900 python scripts/sweep_width_schedule.py --scale 20 --flags 11 --repeats 2 --out $O/sched20_f11.jsonl --schedules '1:768' > $O/sched20_f11.log 2>&1
cat $O/sched17.jsonl $O/sched20.jsonl $O/sched20_f11.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('scale %d flags %d %-22s: %+.2f %% (se %.2f)  sgns %.2f s %s' % (r['scale'], r['flags'], r['schedule'], r['gap_pct'], r['gap_se_pct'], r['sgns_s'], r['waves_and_hot_threshold']))
"
timeout 900 python -m pytest tests/test_n2v_gpu.py -m gpu -x -q > $O/pytest_n2v.log 2>&1
tail -3 $O/pytest_n2v.log
