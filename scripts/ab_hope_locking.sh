set -x
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_hope_gpu.py tests/test_hope_kernels_gpu.py tests/test_lap_gpu.py -x -q 2>&1 | tail -15
timeout 200 python bench.py --workload hope --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/hope_lock_deep.json
GEMHIP_HOPE_FIXED_DEPTH=1 timeout 200 python bench.py --workload hope --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/hope_lock.json
GEMHIP_HOPE_DEBUG=1 timeout 200 python bench.py --workload hope --steps 1 --warmup 0 2>&1 | grep "\[hope\]" | head -60 > gpurun_out/hope_lock_trace.txt
python - <<'PY'
import json
for f in ('hope_lock_deep','hope_lock'):
    try:
        j=json.load(open('gpurun_out/%s.json'%f)); r=j['roofline']; print(f, j['ms_per_step'], r.get('restarts'), r.get('spmm_launches_per_step'), r.get('spmm_seconds_per_step'), r.get('host_eig_seconds_per_step'))
    except Exception as e: print(f, 'ERR', e)
PY
