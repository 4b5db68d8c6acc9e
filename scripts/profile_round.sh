#!/bin/bash
# Round evidence for profiles/: (1) rocprofv3 --kernel-trace --stats over the default bench.py command (all three workloads), (2) PMC traffic
# passes (FETCH_SIZE / WRITE_SIZE, separate runs) over a node2vec run with 2 walks per node and over a GF run.  Output under gpurun_out/prof_r02/.
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_r02; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/all -o all -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.log
db=$(find $out/all -name "*.db" | head -1)
python scripts/rocpd_summary.py "$db" $out/bench_kernel_stats.csv > /dev/null 2>&1
head -12 $out/bench_kernel_stats.csv
rm -rf $out/all
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_n2v_$c -o run -- python bench.py --workload node2vec --num-walks 2 --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_n2v_$c.json 2> $out/pmc_n2v_$c.log
  python scripts/pmc_summary.py $out/pmc_n2v_$c sgns > $out/pmc_n2v_$c.txt; cat $out/pmc_n2v_$c.txt; rm -rf $out/pmc_n2v_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_gf_$c -o run -- python bench.py --workload gf --steps 10 --warmup 2 --no-cpu-baseline > $out/pmc_gf_$c.json 2> $out/pmc_gf_$c.log
  python scripts/pmc_summary.py $out/pmc_gf_$c gf_sweep > $out/pmc_gf_$c.txt; cat $out/pmc_gf_$c.txt; rm -rf $out/pmc_gf_$c
done
