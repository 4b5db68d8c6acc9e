#!/bin/bash
# Round 6, GPU call Q: the two R-MAT scale 22 sequential-oracle embeddings (oracle_push/, 2.1 GB each, ~9 h of CPU each) scored with the GPU evaluator ->
# tests/golden/n2v_ref_oracle_rmat22{,_vocab_order}_e128k.json, and Hogwild launches at the planner's width and at 768 paired with them.
O=gpurun_out/r06q
mkdir -p $O
for fl in 27 11; do
  if [ -f oracle_push/oracle_rmat22_f$fl.npy ]; then
    timeout 1500 python scripts/gpu_score_oracle.py --emb oracle_push/oracle_rmat22_f$fl.npy --scale 22 --flags $fl --out $O --widths 0,768 > $O/score22_f$fl.log 2>&1
    tail -4 $O/score22_f$fl.log | cut -c1-300
  fi
done
ls -la $O
