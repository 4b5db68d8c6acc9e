#!/bin/bash
# Round 6, GPU call L: the tests touched after call K's snapshot (bucket-launch locally hot rows, HOPE verbose with the U side on karate's node order, the
# multi-GPU C-ABI tests with the RCCL loader fix, run_karate on our own driver) and a dry run of scripts/gpu_score_oracle.py on the scale-17 wide-oracle
# embedding (its GPU-scored APs must equal the committed golden's: same seed, same draws).
O=gpurun_out/r06l
mkdir -p $O
timeout 1500 python -m pytest tests/test_n2v_partitioned_gpu.py tests/test_run_sbm_gpu.py tests/test_multi_capi_gpu.py tests/test_run_karate_gpu.py -m gpu -q > $O/pytest_subset.log 2>&1
tail -25 $O/pytest_subset.log | cut -c1-300
timeout 600 python scripts/gpu_score_oracle.py --emb oracle_push/oracle_rmat17_f27_wide.npy --scale 17 --flags 27 --sample 16384 --out $O --widths 0 > $O/score17.log 2>&1
tail -5 $O/score17.log
python - <<'PY'
import json, numpy as np
a = json.load(open('gpurun_out/r06l/n2v_ref_oracle_rmat17_vocab_order_e16k.json')); b = json.load(open('tests/golden/n2v_ref_oracle_rmat17_vocab_order_e16k.json'))
d = np.array(a['ap']) - np.array(b['ap'])
print('GPU-scored wide oracle vs committed (CPU-scored strict oracle) golden: max |dAP| %.2e, MAP %.6f vs %.6f' % (np.abs(d).max(), a['MAP'], b['MAP']))
PY
