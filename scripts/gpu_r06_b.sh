#!/bin/bash
# Round 6, GPU call B: call A refuted stale hub gradients (fresh bits 0/1 cut the hubs' staleness 4x, profiles/r06_staleness_rmat17.jsonl, and moved nothing).
# Next suspect: LOST updates on the mid-frequency rows that carry half of all negative draws (cold by token count, touched by another wavefront inside a
# load..store interval 23-38 % of the time at 768 wavefronts).  fresh bit 2 (value 4): every negative row update an atomic add.  Repeats: a single launch
# at these widths scatters by ~2 %.
set -x
mkdir -p gpurun_out/r06b
O=gpurun_out/r06b
timeout 1200 python scripts/sweep_fresh_hot.py --scale 17 --out $O/fresh_rmat17.jsonl \
  --configs 768:0:27*3,768:4:27*3,768:6:27*3,1536:4:27*2,1536:6:27*2,768:0:27:1000*2,768:4:27:1000*2,768:4:11*2,256:4:27*2 > $O/sweep17.log 2>&1
GEM_HIP_LIB=$PWD/gem_amd/libgem_hip_stale.so GEMHIP_SGNS_STALENESS_OUT=$O/staleness_rmat17.jsonl timeout 600 python scripts/sweep_fresh_hot.py --scale 17 \
  --out $O/fresh_rmat17_stale_build.jsonl --configs 768:4:27,768:6:27 > $O/sweep17_stale.log 2>&1
timeout 1500 python scripts/sweep_fresh_hot.py --scale 20 --out $O/fresh_rmat20.jsonl \
  --configs 768:4:27*2,768:6:27*2,1536:4:27*2,768:4:11 > $O/sweep20.log 2>&1
cat $O/fresh_rmat17.jsonl $O/fresh_rmat20.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('scale %d W %4d fresh %d flags %d hot %d: %+.2f %% (se %.2f)  sgns %.2f s' % (r['scale'], r['max_waves'], r['fresh'], r['flags'], r['hot_count'], r['gap_pct'], r['gap_se_pct'], r['sgns_s']))
"
