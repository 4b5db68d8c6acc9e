#!/bin/bash
# Round 6, GPU call P: what do the hot rows COST in time?  The same pass with count-hot rows switched off (hot-row threshold 0: hubs cached in the LDS windows
# like any row, negatives by reload + store -- the quality is wrong, that is not the question) on R-MAT scale 20 at 256 wavefronts and scale 22 at 548; and the new
# 1M parity test (every GPU seed paired with the oracle's run on the same seed).
set -x
O=gpurun_out/r06p
mkdir -p $O
timeout 900 python scripts/sweep_width_schedule.py --scale 20 --repeats 1 --out $O/hotoff20.jsonl --schedules '1:256:0;1:256;1:768:0' > $O/hotoff20.log 2>&1
timeout 900 python scripts/sweep_width_schedule.py --scale 22 --repeats 1 --out $O/hotoff22.jsonl --schedules '1:548:0;1:548' > $O/hotoff22.log 2>&1
cat $O/hotoff20.jsonl $O/hotoff22.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    if 'schedule' in r: print('scale %d %-10s: %s  sgns %.2f s %s' % (r['scale'], r['schedule'], ('%+.2f %% (se %.2f)' % (r['gap_pct'], r['gap_se_pct'])) if 'gap_pct' in r else 'MAP %.6f' % r['MAP'], r['sgns_s'], r['waves_and_hot_threshold']))
"
timeout 1200 python -m pytest tests/test_bench_gpu.py -m gpu -q -k "headline_size" > $O/pytest_1m.log 2>&1; tail -8 $O/pytest_1m.log | cut -c1-400
