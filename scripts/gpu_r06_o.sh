#!/bin/bash
# Round 6, GPU call O: DEFERRED WRITE-BACK of the re-fetched negative rows in the Hogwild kernels that carry hot rows (sgns.hpp): parity tests of the node2vec
# kernels (single-wavefront Hogwild path == oracle), then time and paired gap on R-MAT scale 17 / 20 / 22 at the planner's width and at 768.
set -x
O=gpurun_out/r06o
mkdir -p $O
timeout 1500 python -m pytest tests/test_n2v_gpu.py tests/test_n2v_partitioned_gpu.py tests/test_edge_cases_gpu.py -m gpu -x -q > $O/pytest_subset.log 2>&1
tail -12 $O/pytest_subset.log | cut -c1-300
timeout 900 python scripts/sweep_width_schedule.py --scale 17 --repeats 2 --out $O/sched17.jsonl --schedules '1:0;1:768' > $O/sched17.log 2>&1
timeout 1500 python scripts/sweep_width_schedule.py --scale 20 --repeats 2 --out $O/sched20.jsonl --schedules '1:0;1:768' > $O/sched20.log 2>&1
timeout 900 python scripts/sweep_width_schedule.py --scale 22 --repeats 1 --out $O/sched22.jsonl --save-ap $O/ap22 --schedules '1:0;1:768' > $O/sched22.log 2>&1
cat $O/sched17.jsonl $O/sched20.jsonl $O/sched22.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    if 'schedule' in r: print('scale %d %-8s: %s  sgns %.2f s %s' % (r['scale'], r['schedule'], ('%+.2f %% (se %.2f)' % (r['gap_pct'], r['gap_se_pct'])) if 'gap_pct' in r else 'MAP %.6f' % r['MAP'], r['sgns_s'], r['waves_and_hot_threshold']))
"
