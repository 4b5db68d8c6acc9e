#!/bin/bash
# Round 6, GPU call G: with the heavy tail gone (call F: launches at one width now agree to 0.2 %) the smooth part of the gap -- -1.6 % at 768 wavefronts on scale
# 17, -3.0 % on scale 20, carried by query nodes of 1 000..30 000 tokens -- can be measured.  Is it the WINDOW copies of warm rows (lower the hot-row threshold:
# fewer concurrent LDS copies) or the staleness of hot rows' gradients (fresh bits, now without the heavy tail's noise)?
set -x
O=gpurun_out/r06g
mkdir -p $O
timeout 1200 python scripts/sweep_width_schedule.py --scale 17 --repeats 2 --out $O/hot17.jsonl --schedules '1:768:1900;1:768:950;1:768:480;1:768:240;1:1536:480' > $O/hot17.log 2>&1
timeout 600 python scripts/sweep_width_schedule.py --scale 17 --repeats 2 --fresh 3 --out $O/fresh17.jsonl --schedules '1:768;1:768:480' > $O/fresh17.log 2>&1
timeout 1500 python scripts/sweep_width_schedule.py --scale 20 --repeats 2 --out $O/hot20.jsonl --schedules '1:768:13400;1:768:6700;1:768:3350;1:768:1675;1:1536:1675' > $O/hot20.log 2>&1
timeout 900 python scripts/sweep_width_schedule.py --scale 20 --repeats 2 --fresh 3 --out $O/fresh20.jsonl --schedules '1:768;1:768:3350' > $O/fresh20.log 2>&1
cat $O/hot17.jsonl $O/fresh17.jsonl $O/hot20.jsonl $O/fresh20.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    if 'schedule' in r: print('scale %d fresh %d %-14s: %+.2f %% (se %.2f)  sgns %.2f s %s' % (r['scale'], r['fresh'], r['schedule'], r['gap_pct'], r['gap_se_pct'], r['sgns_s'], r['waves_and_hot_threshold']))
"
