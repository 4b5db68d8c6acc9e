#!/bin/bash
# Round-4 evidence for profiles/ (output under gpurun_out/prof_r04/):
#  bench   the default `python bench.py` line, un-profiled (one run, nothing stitched)
#  trace   rocprofv3 --kernel-trace --stats over one node2vec pass and over the gf / hope workloads (same commands as the lines beside them)
#  pmc     FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only): a GF run (gf_sweep_rows_kernel) and a node2vec run with 2 walks per node
#          (calibration factors of this access pattern: profiles/r03_pmc_traffic.json, reproduced on two boxes)
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_r04; mkdir -p $out
what=${1:-all}
if [ "$what" = all ] || [ "$what" = bench ]; then
  python bench.py > $out/bench_all.json 2> $out/bench_all.log; tail -c 400 $out/bench_all.json; echo
fi
if [ "$what" = all ] || [ "$what" = trace ]; then
  for wl in node2vec gf hope; do
    extra="--steps 1 --warmup 0"; [ $wl = gf ] && extra="--steps 50 --warmup 5"; [ $wl = hope ] && extra="--steps 5 --warmup 1"
    rocprofv3 --kernel-trace --stats -d $out/tr_$wl -o $wl -- python bench.py --workload $wl $extra --no-cpu-baseline --no-api-wall > $out/bench_${wl}_under_rocprof.json 2> $out/bench_${wl}_under_rocprof.log
    db=$(find $out/tr_$wl -name "*.db" | head -1)
    python scripts/rocpd_summary.py "$db" $out/bench_${wl}_kernel_stats.csv > /dev/null 2>&1
    head -6 $out/bench_${wl}_kernel_stats.csv
    rm -rf $out/tr_$wl
  done
fi
if [ "$what" = all ] || [ "$what" = pmc ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_gf_$c -o run -- python bench.py --workload gf --steps 10 --warmup 2 --no-cpu-baseline --no-api-wall > $out/pmc_gf_$c.json 2> $out/pmc_gf_$c.log
    python scripts/pmc_summary.py $out/pmc_gf_$c gf_sweep > $out/pmc_gf_$c.txt; cat $out/pmc_gf_$c.txt; rm -rf $out/pmc_gf_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_n2v_$c -o run -- python bench.py --workload node2vec --num-walks 2 --steps 1 --warmup 0 --no-cpu-baseline --no-api-wall > $out/pmc_n2v_$c.json 2> $out/pmc_n2v_$c.log
    python scripts/pmc_summary.py $out/pmc_n2v_$c sgns > $out/pmc_n2v_$c.txt; cat $out/pmc_n2v_$c.txt; rm -rf $out/pmc_n2v_$c
  done
fi
