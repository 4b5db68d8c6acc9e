# A/B of the HOPE solver knobs on the GPU box: basis size of the deep phase, partial vs full
# projected eigensolver.  Usage: gpurun -- 'bash scripts/hope_basis_sweep.sh'
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --workload hope --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/hope_$tag.json; }
run full_320 GEMHIP_EIG_FULL=1
for c in 320 384 448 512; do run top_$c GEMHIP_HOPE_BASIS_COLS=$c; done
python - <<'PY'
import json
for t in ('full_320', 'top_320', 'top_384', 'top_448', 'top_512'):
    try:
        j = json.load(open('gpurun_out/hope_%s.json' % t)); r = j['roofline']
        print(t, round(j['ms_per_step'], 1), 'ms  cycles', r.get('restarts'), 'spmm', r.get('spmm_launches_per_step'), round(r.get('spmm_seconds_per_step'), 4), 's  eig', round(r.get('host_eig_seconds_per_step'), 4), 's')
    except Exception as e:
        print(t, 'ERR', e)
PY
