#!/bin/bash
# Round 6 evidence (output under gpurun_out/prof_r06/):
#  bench   the default `python bench.py` line, un-profiled (one run, nothing stitched): stdout (the compact line) and stderr (BENCH_DETAIL + log) kept apart
#  trace   rocprofv3 --kernel-trace --stats over one node2vec pass and over the gf / hope workloads (same commands as the lines beside them)
#  pmc-rmat  FETCH_SIZE / WRITE_SIZE of the SGNS kernel on R-MAT scale 22 (bytes per pair on the power-law graph), separate passes
#  pmc-hope  FETCH_SIZE / WRITE_SIZE of the SpMM kernel over eigen-path solves
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_r06; mkdir -p $out
what=${1:-all}
if [ "$what" = all ] || [ "$what" = bench ]; then
  python bench.py > $out/bench_stdout.txt 2> $out/bench_stderr.txt; wc -c $out/bench_stdout.txt; cat $out/bench_stdout.txt; echo
  grep '^BENCH_DETAIL ' $out/bench_stderr.txt | sed 's/^BENCH_DETAIL //' > $out/bench_detail.json
fi
if [ "$what" = all ] || [ "$what" = trace ]; then
  for wl in node2vec gf hope; do
    extra="--steps 1 --warmup 0"; [ $wl = gf ] && extra="--steps 50 --warmup 5"; [ $wl = hope ] && extra="--steps 5 --warmup 1"
    timeout 600 rocprofv3 --kernel-trace --stats -d $out/tr_$wl -o $wl -- python bench.py --workload $wl $extra --no-cpu-baseline --no-api-wall > $out/bench_${wl}_under_rocprof.json 2> $out/bench_${wl}_under_rocprof.log
    db=$(find $out/tr_$wl -name "*.db" | head -1)
    python scripts/rocpd_summary.py "$db" $out/bench_${wl}_kernel_stats.csv > /dev/null 2>&1
    head -4 $out/bench_${wl}_kernel_stats.csv | cut -c1-200
    rm -rf $out/tr_$wl
  done
fi
if [ "$what" = all ] || [ "$what" = pmc-sbm ]; then
  # headline kernel: FETCH_SIZE / WRITE_SIZE over one full SBM 1M/10M pass (the default launch: 1792 wavefronts), separate passes
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_sbm_$c -o run -- python bench.py --workload node2vec --steps 1 --warmup 0 --no-cpu-baseline --no-api-wall > $out/pmc_sbm_$c.json 2> $out/pmc_sbm_$c.log
    python scripts/pmc_summary.py $out/pmc_sbm_$c sgns_win > $out/pmc_sbm_$c.txt; cat $out/pmc_sbm_$c.txt; rm -rf $out/pmc_sbm_$c
  done
fi
if [ "$what" = all ] || [ "$what" = microbench ]; then
  scripts/microbench/northstar > $out/microbench_northstar.jsonl 2>&1; cat $out/microbench_northstar.jsonl
fi
if [ "$what" = all ] || [ "$what" = pmc-rmat ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_rmat_$c -o run -- python bench.py --workload node2vec --graph rmat --nodes 4194304 --edges 64000000 --steps 1 --warmup 0 --no-cpu-baseline --no-api-wall > $out/pmc_rmat_$c.json 2> $out/pmc_rmat_$c.log
    python scripts/pmc_summary.py $out/pmc_rmat_$c sgns_win > $out/pmc_rmat_$c.txt; cat $out/pmc_rmat_$c.txt; rm -rf $out/pmc_rmat_$c
  done
fi
if [ "$what" = all ] || [ "$what" = pmc-hope ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_hope_$c -o run -- python bench.py --workload hope --steps 5 --warmup 1 --no-cpu-baseline --no-api-wall > $out/pmc_hope_$c.json 2> $out/pmc_hope_$c.log
    python scripts/pmc_summary.py $out/pmc_hope_$c hope_spmm16 > $out/pmc_hope_$c.txt; cat $out/pmc_hope_$c.txt; rm -rf $out/pmc_hope_$c
  done
fi
