#!/bin/bash
# Round 6, GPU call R: the closing tier on the final library -- the whole GPU test suite (now with the scale-22 parity test) and the default bench line.
O=gpurun_out/r06r
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -4 $O/smoke.log
timeout 3000 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu_full.log 2>&1; tail -30 $O/pytest_gpu_full.log | cut -c1-300
( time python bench.py > $O/bench_stdout.txt 2> $O/bench_stderr.txt ) 2> $O/bench_time.txt; cat $O/bench_time.txt; wc -c $O/bench_stdout.txt; cat $O/bench_stdout.txt | cut -c1-4000
grep '^BENCH_DETAIL ' $O/bench_stderr.txt | sed 's/^BENCH_DETAIL //' > $O/bench_detail.json
