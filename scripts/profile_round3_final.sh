#!/bin/bash
# Round-3 closing evidence after the seven-wavefronts-per-CU change (output under gpurun_out/prof_r03b/): the default `python bench.py` line,
# rocprofv3 --kernel-trace --stats of one node2vec step, and the FETCH_SIZE / WRITE_SIZE passes of a 2-walks-per-node node2vec run
# (calibration factors: profiles/r03_pmc_traffic.json, scripts/profile_round3.sh pmc).
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_r03b; mkdir -p $out
python bench.py > $out/bench_all.json 2> $out/bench_all.log; tail -c 300 $out/bench_all.json; echo
rocprofv3 --kernel-trace --stats -d $out/tr_node2vec -o node2vec -- python bench.py --workload node2vec --steps 1 --warmup 0 --no-cpu-baseline > $out/bench_node2vec_under_rocprof.json 2> $out/bench_node2vec_under_rocprof.log
db=$(find $out/tr_node2vec -name "*.db" | head -1)
python scripts/rocpd_summary.py "$db" $out/bench_node2vec_kernel_stats.csv > /dev/null 2>&1
head -5 $out/bench_node2vec_kernel_stats.csv
rm -rf $out/tr_node2vec
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_n2v_$c -o run -- python bench.py --workload node2vec --num-walks 2 --steps 1 --warmup 0 --no-cpu-baseline > $out/pmc_n2v_$c.json 2> $out/pmc_n2v_$c.log
  python scripts/pmc_summary.py $out/pmc_n2v_$c sgns > $out/pmc_n2v_$c.txt; cat $out/pmc_n2v_$c.txt; rm -rf $out/pmc_n2v_$c
done
