#!/bin/bash
# Round 6, GPU call A: FRESH HOT ROWS (sgns.hpp, SgnsArgs::fresh) against the sequential oracle on R-MAT scale 17 and 20, widths x fresh bits, and the
# staleness histograms of the instrumented build on scale 17; then the node2vec GPU tests on the new library.
set -x
mkdir -p gpurun_out/r06a
O=gpurun_out/r06a
timeout 900 python scripts/sweep_fresh_hot.py --scale 17 --out $O/fresh_rmat17.jsonl \
  --configs 256:0:27,768:0:27,768:1:27,768:2:27,768:3:27,1536:0:27,1536:3:27,768:0:11,768:3:11,1536:3:11 > $O/sweep17.log 2>&1
GEM_HIP_LIB=$PWD/gem_amd/libgem_hip_stale.so GEMHIP_SGNS_STALENESS_OUT=$O/staleness_rmat17.jsonl timeout 600 python scripts/sweep_fresh_hot.py --scale 17 \
  --out $O/fresh_rmat17_stale_build.jsonl --configs 256:0:27,768:0:27,768:3:27,1536:3:27 > $O/sweep17_stale.log 2>&1
timeout 1500 python scripts/sweep_fresh_hot.py --scale 20 --out $O/fresh_rmat20.jsonl \
  --configs 768:0:27,768:3:27,1536:3:27,768:1:27,768:2:27,207:3:27,768:3:11,768:0:11 > $O/sweep20.log 2>&1
timeout 900 python -m pytest tests/test_n2v_gpu.py -m gpu -x -q > $O/pytest_n2v.log 2>&1
tail -5 $O/pytest_n2v.log
cat $O/fresh_rmat17.jsonl $O/fresh_rmat20.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('scale %d W %4d fresh %d flags %d: %+.2f %% (se %.2f)  sgns %.2f s' % (r['scale'], r['max_waves'], r['fresh'], r['flags'], r['gap_pct'], r['gap_se_pct'], r['sgns_s']))
"
