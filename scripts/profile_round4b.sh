#!/bin/bash
# Round 4, closing evidence (second session; output under gpurun_out/prof_r04b/):
#  pmc-gf  FETCH_SIZE / WRITE_SIZE of the GF sweep (own-row loads non-temporal now) -> profiles/r04b_pmc_traffic.json is rebuilt BEFORE the bench runs,
#          because bench.py replays the newest committed per-unit traffic figure into roofline.traffic
#  tests   the full GPU tier
#  bench   the default `python bench.py` line, un-profiled (one run, nothing stitched)
#  trace   rocprofv3 --kernel-trace --stats over one node2vec pass and over the gf / hope workloads (same commands as the lines beside them)
#  sq      SQ instruction counters of the SGNS kernel (one walk per node)
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_r04b; mkdir -p $out
what=${1:-all}
if [ "$what" = all ] || [ "$what" = pmc-gf ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_gf_$c -o run -- python bench.py --workload gf --steps 10 --warmup 2 --no-cpu-baseline --no-api-wall > $out/pmc_gf_$c.json 2> $out/pmc_gf_$c.log
    python scripts/pmc_summary.py $out/pmc_gf_$c gf_sweep > $out/pmc_gf_$c.txt; cat $out/pmc_gf_$c.txt; rm -rf $out/pmc_gf_$c
  done
  python scripts/make_pmc_traffic.py $out $out/r04b_pmc_traffic.json && cp $out/r04b_pmc_traffic.json profiles/r04b_pmc_traffic.json
fi
if [ "$what" = all ] || [ "$what" = tests ]; then
  ( timeout 1200 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -30 ) > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
fi
if [ "$what" = all ] || [ "$what" = bench ]; then
  python bench.py > $out/bench_all.json 2> $out/bench_all.log; tail -c 300 $out/bench_all.json; echo
fi
if [ "$what" = all ] || [ "$what" = trace ]; then
  for wl in node2vec gf hope; do
    extra="--steps 1 --warmup 0"; [ $wl = gf ] && extra="--steps 50 --warmup 5"; [ $wl = hope ] && extra="--steps 5 --warmup 1"
    timeout 600 rocprofv3 --kernel-trace --stats -d $out/tr_$wl -o $wl -- python bench.py --workload $wl $extra --no-cpu-baseline --no-api-wall > $out/bench_${wl}_under_rocprof.json 2> $out/bench_${wl}_under_rocprof.log
    db=$(find $out/tr_$wl -name "*.db" | head -1)
    python scripts/rocpd_summary.py "$db" $out/bench_${wl}_kernel_stats.csv > /dev/null 2>&1
    head -5 $out/bench_${wl}_kernel_stats.csv | cut -c1-200
    rm -rf $out/tr_$wl
  done
fi
if [ "$what" = all ] || [ "$what" = sq ]; then
  PMC_GROUPS="0 2" bash scripts/pmc_passes.sh r04b_sgns sgns_win -- python bench.py --workload node2vec --num-walks 1 --steps 1 --warmup 0 --no-cpu-baseline --no-api-wall
  mkdir -p $out/pmc_sq; cp gpurun_out/pmc/r04b_sgns_g*.txt $out/pmc_sq/ 2>/dev/null
fi
