#!/usr/bin/env python3
"""Pair the per-node APs the round-6 GPU sweeps saved on R-MAT scale 22 (131 072-node eligible sample) with the oracle goldens -- over whichever sample the golden
holds (the 16 384-node sub-sample is a subset of the 131 072: same RandomState, a prefix of the draw) -> profiles/r06_rmat22_paired.jsonl."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gem_amd.graph import rmat_graph
from gem_amd.evaluation import reconstruction as gr
g = rmat_graph(22, 64000000, 20260928)
big = np.asarray(gr.eligible_sample(g, 131072))
RUNS = {27: [('r06u/ap22/ap_scale22_f27_s1_r0.npy', '128'), ('r06h/ap22/ap_scale22_f27_s1_r0.npy', '256 (library before the zero-update skip)'),
             ('r06u/ap22/ap_scale22_f27_s0_r0.npy', '332'), ('r06w/ap22/ap_scale22_f27_s0_r0.npy', '548 (the planner, final library)'),
             ('r06h/ap22/ap_scale22_f27_s0_r0.npy', '548 (library before the zero-update skip)'), ('r06w/ap22/ap_scale22_f27_s1_r0.npy', '768 (final library)'),
             ('r06h/ap22/ap_scale22_f27_s2_r0.npy', '768 (library before the zero-update skip)')],
        11: [('r06u/ap22/ap_scale22_f11_s0_r0.npy', '332'), ('r06w/ap22/ap_scale22_f11_s0_r0.npy', '548 (the planner, final library)'),
             ('r06h/ap22/ap_scale22_f11_s0_r0.npy', '548 (library before the zero-update skip)')]}
out = open(os.path.join(ROOT, 'profiles', 'r06_rmat22_paired.jsonl'), 'w')
for fl, name in ((27, 'n2v_ref_oracle_rmat22_vocab_order'), (11, 'n2v_ref_oracle_rmat22')):
    gp = None
    for suf in ('e128k', 'e16k'):
        p = os.path.join(ROOT, 'tests', 'golden', '%s_%s.json' % (name, suf))
        if gp is None and os.path.exists(p):
            gp = p
    if gp is None:
        continue
    ref = json.load(open(gp)); apo = np.asarray(ref['ap'])
    nodes = np.asarray(gr.eligible_sample(g, len(apo)))
    idx = np.searchsorted(big, nodes); assert np.array_equal(big[idx], nodes)
    for f, w in RUNS[fl]:
        fp = os.path.join(ROOT, 'gpurun_out', f)
        if not os.path.exists(fp):
            continue
        ap = np.load(fp)[idx]; dd = ap - apo
        rec = {'flags': fl, 'wavefronts': w, 'golden': os.path.relpath(gp, ROOT), 'nodes': int(len(apo)), 'MAP': float(ap.mean()), 'oracle_MAP': float(apo.mean()),
               'gap_pct': float(100 * dd.mean() / apo.mean()), 'gap_se_pct': float(100 * dd.std(ddof=1) / np.sqrt(len(dd)) / apo.mean())}
        s = json.dumps(rec); print(s); out.write(s + '\n')
