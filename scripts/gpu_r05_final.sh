#!/bin/bash
# Round 5, closing GPU call: the whole GPU tier exactly as the driver runs it (-x), smoke(), then the default bench line (un-profiled).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_r05_final; mkdir -p $out
python -m pytest tests -x -q -m gpu --durations=8 > $out/pytest_gpu.log 2>&1; tail -25 $out/pytest_gpu.log | cut -c1-260
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -3 $out/smoke.log
python bench.py > $out/bench_stdout.txt 2> $out/bench_stderr.txt; wc -c $out/bench_stdout.txt; cut -c1-600 $out/bench_stdout.txt
grep '^BENCH_DETAIL ' $out/bench_stderr.txt | sed 's/^BENCH_DETAIL //' > $out/bench_detail.json
