#!/bin/bash
# Round 6, GPU call M: call G lowered the hot-row threshold (more rows on the atomic path) and the gap grew.  The other direction: FEWER hot rows (threshold x2, x4,
# x8: more rows cached in the LDS windows, fewer memory-side atomics) at 768 wavefronts, against the oracle; and the time of an R-MAT-22 pass with exactly-zero
# hot updates skipped.
set -x
O=gpurun_out/r06m
mkdir -p $O
timeout 900 python scripts/sweep_width_schedule.py --scale 17 --repeats 2 --out $O/hot17.jsonl --schedules '1:768:7636;1:768:15272;1:768:30544;1:768' > $O/hot17.log 2>&1
timeout 1500 python scripts/sweep_width_schedule.py --scale 20 --repeats 2 --out $O/hot20.jsonl --schedules '1:768:53630;1:768:107260;1:768:214520;1:768' > $O/hot20.log 2>&1
timeout 900 python scripts/sweep_width_schedule.py --scale 22 --repeats 1 --out $O/sched22.jsonl --schedules '1:0' > $O/sched22.log 2>&1
cat $O/hot17.jsonl $O/hot20.jsonl $O/sched22.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    if 'schedule' in r: print('scale %d %-16s: %s  sgns %.2f s %s' % (r['scale'], r['schedule'], ('%+.2f %% (se %.2f)' % (r['gap_pct'], r['gap_se_pct'])) if 'gap_pct' in r else 'MAP %.6f' % r['MAP'], r['sgns_s'], r['waves_and_hot_threshold']))
"
