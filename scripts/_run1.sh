mkdir -p gpurun_out/r02
python scripts/ab_hope_sym.py > gpurun_out/r02/ab_hope_sym.jsonl 2> gpurun_out/r02/ab_hope_sym.err
tail -5 gpurun_out/r02/ab_hope_sym.err
cat gpurun_out/r02/ab_hope_sym.jsonl | cut -c1-420
timeout 400 python -m pytest tests/test_hope_gpu.py tests/test_edge_cases_gpu.py -m gpu -x -q 2>&1 | tail -8
