#!/usr/bin/env python3
"""Pair per-node AP arrays a GPU sweep saved (scripts/sweep_width_schedule.py --save-ap, same node sample: reconstruction.eligible_sample) with an oracle golden
that was made later:  python scripts/pair_saved_aps.py tests/golden/n2v_ref_oracle_rmat22_vocab_order_e128k.json gpurun_out/r06h/ap22/ap_scale22_f27_s*_r0.npy"""
import json, sys
import numpy as np
ref = json.load(open(sys.argv[1]))
apo = np.asarray(ref['ap'])
for f in sys.argv[2:]:
    ap = np.load(f)
    assert len(ap) == len(apo), (f, len(ap), len(apo))
    dd = ap - apo
    print(json.dumps({'file': f, 'golden': sys.argv[1], 'MAP': float(ap.mean()), 'oracle_MAP': float(apo.mean()), 'gap_pct': float(100 * dd.mean() / apo.mean()),
                      'gap_se_pct': float(100 * dd.std(ddof=1) / np.sqrt(len(dd)) / apo.mean())}))
