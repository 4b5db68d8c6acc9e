#!/bin/bash
# Round 5, GPU call G: HOPE SpMM variants (scalar columns / 16-byte accesses at full and half gather depth): bit-identity test, in-process A/B on the
# eigen-path solve, separate-process A/B on the directed solve.
python -m pytest tests/test_hope_kernels_gpu.py tests/test_hope_gpu.py -m gpu -q 2>&1 | tail -4
AB_SPMM_VARIANTS=0,1,2,0,1,2 python scripts/ab_hope_sym.py 2> gpurun_out/r05_ab_hope_spmm16v.err | grep sym_spmm > gpurun_out/r05_ab_hope_spmm16v.jsonl
for v in 0 1 0 1; do GEMHIP_HOPE_SPMM16V=$v python bench.py --workload hope --hope-directed --no-cpu-baseline --no-api-wall --steps 3 --warmup 1 2>&1 >/dev/null | grep '^BENCH_DETAIL ' | sed "s/^BENCH_DETAIL /{\"spmm16v\": $v, \"line\": /; s/$/}/" >> gpurun_out/r05_ab_hope_spmm16v_directed.jsonl; done
python - <<'PY'
import json
for l in open('gpurun_out/r05_ab_hope_spmm16v.jsonl'):
    j = json.loads(l); print(j['variant'], round(j['seconds_min'] * 1e3, 3), 'ms; spmm', round(j['spmm_seconds'] * 1e3, 3), 'ms', j['max_rel_err_vs_arpack'])
for l in open('gpurun_out/r05_ab_hope_spmm16v_directed.jsonl'):
    j = json.loads(l); print('directed spmm16v', j['spmm16v'], j['line']['ms_per_step'], j['line']['roofline']['avg_launch_us'])
PY
tail -3 gpurun_out/r05_ab_hope_spmm16v.err
