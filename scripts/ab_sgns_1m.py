#!/usr/bin/env python3
"""A/B of sgns_win_kernel's Hogwild settings at the headline size (SBM 1M/10M, d=128): concurrent wavefronts x pairs of negative rows
requested ahead x reload-on-update (gemhip_sgns_set_hogwild).  Seconds of the SGNS launch and the reconstruction MAP over the sample of
tests/golden/n2v_ref_oracle_1000k.json, PAIRED with the sequential oracle's run (same seed -> same walks, same negatives).
    python scripts/ab_sgns_1m.py [out.jsonl] [waves:prefetch:reload ...]"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from gem_amd import _hip
from gem_amd.graph import edge_arrays, sbm_graph
from gem_amd.evaluation import reconstruction as gr
from test_n2v_gpu import Dev

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'ab_sgns_1m.jsonl')
cfgs = sys.argv[2:] or ['1024:2:0', '1024:2:1', '1536:2:1', '1536:1:1', '1536:2:0']
ref = json.load(open(os.path.join(ROOT, 'tests', 'golden', os.environ.get('GEM_AB_GOLDEN', 'n2v_ref_oracle_1000k.json'))))     # (GEM_AB_GOLDEN=n2v_ref_oracle_1000k_s4096.json: the 4096-node sample)
pr = ref['params']
g = sbm_graph(pr['n'], pr['edges'], pr['blocks'], pr['seed'])
nodes = np.random.RandomState(0).choice(g.n, size=len(ref['ap']), replace=False)
n, src, dst, w, _ = edge_arrays(g)
dev = Dev(n, src, dst, w)
SEED = 20260923
m = C.c_int64(); _hip.check(dev.L.gemhip_n2v_start_nodes(dev.h, C.byref(m)))
_hip.check(dev.L.gemhip_n2v_walks(dev.h, 1.0, 1.0, pr['num_walks'], pr['walk_len'], SEED, 11, 0, m.value * pr['num_walks'], None))
dev.unigram()
tot = m.value * pr['num_walks'] * pr['walk_len']
P = np.empty((n, pr['d']), np.float32)
os.makedirs(os.path.dirname(out), exist_ok=True)
for cs in cfgs:
    waves, pf, rl = (int(x) for x in cs.split(':'))
    _hip.check(dev.L.gemhip_sgns_set_hogwild(dev.h, pf, rl))
    _hip.check(dev.L.gemhip_n2v_set_max_waves(dev.h, waves))
    _hip.check(dev.L.gemhip_sgns_init(dev.h, pr['d'], SEED, None, None))
    _hip.check(dev.L.gemhip_synchronize(None))
    pairs = C.c_int64(); _hip.check(dev.L.gemhip_sgns_pairs(dev.h, C.byref(pairs), 1))
    t = time.time()
    _hip.check(dev.L.gemhip_sgns_train(dev.h, pr['window'], 5, 0.025, 1, 0, 0, m.value * pr['num_walks'], tot, 0, SEED, 11, None))
    _hip.check(dev.L.gemhip_synchronize(None))
    el = time.time() - t
    _hip.check(dev.L.gemhip_sgns_pairs(dev.h, C.byref(pairs), 0))
    _hip.check(dev.L.gemhip_sgns_get_tables(dev.h, _hip.ptr(P, C.c_float), None))
    ap = gr.sampled_ap_gpu(g, None, P, nodes)
    dd = ap - np.asarray(ref['ap'])
    rec = dict(cfg=cs, waves=waves, prefetch=pf, reload=rl, seconds=el, pairs=pairs.value, algorithmic_TBs=pairs.value * 7192 / el / 1e12, MAP=float(ap.mean()),
               oracle_MAP=ref['MAP'], rel_pct=float(100 * dd.mean() / ref['MAP']), rel_se_pct=float(100 * dd.std(ddof=1) / np.sqrt(len(dd)) / ref['MAP']),
               finite=bool(np.isfinite(P).all()))
    print(json.dumps(rec), flush=True)
    open(out, 'a').write(json.dumps(rec) + '\n')
dev.close()
