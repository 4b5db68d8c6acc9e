#!/bin/bash
cd /root/repo
: > profiles/r06_rmat22_paired.jsonl
G27=tests/golden/n2v_ref_oracle_rmat22_vocab_order_e128k.json; G11=tests/golden/n2v_ref_oracle_rmat22_e128k.json
if [ -f $G27 ]; then
  for spec in "r06u/ap22/ap_scale22_f27_s1_r0.npy 128" "r06h/ap22/ap_scale22_f27_s1_r0.npy 256(pre-zero-skip-library)" "r06u/ap22/ap_scale22_f27_s0_r0.npy 332" "r06w/ap22/ap_scale22_f27_s0_r0.npy 548(planner,final-library)" "r06h/ap22/ap_scale22_f27_s0_r0.npy 548(pre-zero-skip-library)" "r06w/ap22/ap_scale22_f27_s1_r0.npy 768(final-library)" "r06h/ap22/ap_scale22_f27_s2_r0.npy 768(pre-zero-skip-library)"; do
    set -- $spec; python scripts/pair_saved_aps.py $G27 gpurun_out/$1 | sed "s/^{/{\"flags\": 27, \"wavefronts\": \"$2\", /" >> profiles/r06_rmat22_paired.jsonl; done
fi
if [ -f $G11 ]; then
  for spec in "r06u/ap22/ap_scale22_f11_s0_r0.npy 332" "r06w/ap22/ap_scale22_f11_s0_r0.npy 548(planner,final-library)" "r06h/ap22/ap_scale22_f11_s0_r0.npy 548(pre-zero-skip-library)"; do
    set -- $spec; python scripts/pair_saved_aps.py $G11 gpurun_out/$1 | sed "s/^{/{\"flags\": 11, \"wavefronts\": \"$2\", /" >> profiles/r06_rmat22_paired.jsonl; done
fi
python - <<'PY'
import json
for l in open('profiles/r06_rmat22_paired.jsonl'):
    r=json.loads(l); print('flags %d W %-32s: %+.2f %% (se %.2f)  MAP %.6f vs oracle %.6f'%(r['flags'],r['wavefronts'],r['gap_pct'],r['gap_se_pct'],r['MAP'],r['oracle_MAP']))
PY
