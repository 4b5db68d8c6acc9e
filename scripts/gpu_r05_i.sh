#!/bin/bash
# Round 5, GPU call I: A/B of the compile-time row width in the FULL instantiations of sgns_win_kernel (libgem_hip_rtd.so = runtime d, the round-4 form):
# SBM 1M/10M full passes and R-MAT scale 22 (2 walks per node), libraries alternated.
for rep in 1 2; do
  for lib in rtd new; do
    L=""; [ $lib = rtd ] && L=$PWD/gem_amd/libgem_hip_rtd.so
    GEM_HIP_LIB=$L python bench.py --workload node2vec --steps 1 --warmup 0 --no-cpu-baseline --no-api-wall 2>&1 >/dev/null | grep '^BENCH_DETAIL ' | sed "s/^BENCH_DETAIL /{\"lib\": \"$lib\", \"graph\": \"sbm1m\", \"line\": /; s/$/}/" >> gpurun_out/r05_ab_sgns_const_d.jsonl
  done
done
for lib in rtd new rtd new; do
  L=""; [ $lib = rtd ] && L=$PWD/gem_amd/libgem_hip_rtd.so
  GEM_HIP_LIB=$L python bench.py --workload node2vec --graph rmat --nodes 4194304 --edges 64000000 --num-walks 3 --steps 1 --warmup 0 --no-cpu-baseline --no-api-wall 2>&1 >/dev/null | grep '^BENCH_DETAIL ' | sed "s/^BENCH_DETAIL /{\"lib\": \"$lib\", \"graph\": \"rmat22_r3\", \"line\": /; s/$/}/" >> gpurun_out/r05_ab_sgns_const_d.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r05_ab_sgns_const_d.jsonl'):
    j = json.loads(l); r = j['line']['roofline']; q = j['line'].get('quality', {})
    print(j['lib'], j['graph'], 'sgns launch %.3f s' % (r['avg_launch_us'] / 1e6), 'frac %.4f' % r['frac'], 'map-oracle', q.get('map_minus_oracle_map'))
PY
