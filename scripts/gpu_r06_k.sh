#!/bin/bash
# Round 6, GPU call K: the closing tier -- smoke, the whole GPU test suite (the driver's command), the round's profile script (default bench line, rocprofv3
# kernel traces, PMC passes, the north_star microbenchmarks).
O=gpurun_out/r06k
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -4 $O/smoke.log
timeout 300 python bench.py --workload node2vec --graph rmat --nodes 131072 --edges 2000000 --steps 1 --warmup 0 --no-cpu-baseline --no-api-wall > $O/bench_small_rmat17.json 2> $O/bench_small_rmat17.log; tail -c 1500 $O/bench_small_rmat17.json; echo
timeout 2700 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu_full.log 2>&1; tail -40 $O/pytest_gpu_full.log | cut -c1-300
bash scripts/profile_round6.sh all > $O/profile_round6.log 2>&1; tail -60 $O/profile_round6.log | cut -c1-400
mkdir -p $O/prof; cp -r gpurun_out/prof_r06/* $O/prof/ 2>/dev/null
