#!/bin/bash
# round 4, GPU call C: the whole GPU tier, then the round's bench line, traces and PMC passes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04c; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/ -q -m gpu 2>&1 | tail -40 ) > $O/pytest_gpu.log 2>&1
tail -6 $O/pytest_gpu.log
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -3 $O/smoke.log
bash scripts/profile_round4.sh all 2>&1 | tail -60
