cd $GRAFT_REPO_ROOT
for c in 320 384 448 512; do
GEMHIP_HOPE_BASIS_COLS=$c timeout 200 python bench.py --workload hope --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/hope_basis_$c.json
done
python - <<'PY'
import json
for c in (320,384,448,512):
    try:
        j=json.load(open('gpurun_out/hope_basis_%d.json'%c)); r=j['roofline']; print(c, round(j['ms_per_step'],1), r.get('restarts'), r.get('spmm_launches_per_step'), round(r.get('spmm_seconds_per_step'),4), round(r.get('host_eig_seconds_per_step'),4))
    except Exception as e: print(c, 'ERR', e)
PY
