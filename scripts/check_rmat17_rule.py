#!/usr/bin/env python3
"""The Hogwild concurrency rule on a second graph family: R-MAT scale 17 (131 072 nodes, 1.86 M edges, power-law hubs) against the
sequential oracle's run on the same seed (tests/golden/n2v_ref_oracle_rmat17.json, scripts/make_golden_n2v_scale.py --rmat-scale 17 --engine oracle):
paired per-node AP gap at the default (reload-on-update, rho <= 1.5 % -> n/133 wavefronts), without reload at its own cap (n/1000) and at the
round-2 cap (1024 wavefronts without reload: rho = 11 % here)."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from gem_amd import _hip
from gem_amd.graph import edge_arrays, rmat_graph
from gem_amd.evaluation import reconstruction as gr
from test_n2v_gpu import Dev
ref = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'n2v_ref_oracle_rmat17.json')))
pr = ref['params']
g = rmat_graph(pr['rmat_scale'], pr['edges'], pr['seed'])
nodes = np.random.RandomState(0).choice(g.n, size=len(ref['ap']), replace=False)
n, src, dst, w, _ = edge_arrays(g)
dev = Dev(n, src, dst, w)
SEED = 20260923
m = C.c_int64(); _hip.check(dev.L.gemhip_n2v_start_nodes(dev.h, C.byref(m)))
_hip.check(dev.L.gemhip_n2v_walks(dev.h, 1.0, 1.0, pr['num_walks'], pr['walk_len'], SEED, 11, 0, m.value * pr['num_walks'], None))
dev.unigram()
tot = m.value * pr['num_walks'] * pr['walk_len']
P = np.empty((n, pr['d']), np.float32)
for name, waves, pf, rl in (('default', 0, 2, 1), ('default', 0, 2, 1), ('no_reload_own_cap', 0, 2, 0), ('no_reload_1024_waves', 1024, 2, 0), ('reload_1536_waves', 1536, 2, 1)):
    _hip.check(dev.L.gemhip_sgns_set_hogwild(dev.h, pf, rl))
    _hip.check(dev.L.gemhip_n2v_set_max_waves(dev.h, waves))
    _hip.check(dev.L.gemhip_sgns_init(dev.h, pr['d'], SEED, None, None))
    _hip.check(dev.L.gemhip_synchronize(None))
    t = time.time()
    _hip.check(dev.L.gemhip_sgns_train(dev.h, pr['window'], 5, 0.025, 1, 0, 0, m.value * pr['num_walks'], tot, 0, SEED, 11, None))
    _hip.check(dev.L.gemhip_synchronize(None))
    el = time.time() - t
    _hip.check(dev.L.gemhip_sgns_get_tables(dev.h, _hip.ptr(P, C.c_float), None))
    ap = gr.sampled_ap_gpu(g, None, P, nodes)
    dd = ap - np.asarray(ref['ap'])
    print(json.dumps(dict(cfg=name, waves=waves, reload=rl, seconds=round(el, 3), MAP=float(ap.mean()), oracle_MAP=ref['MAP'], rel_pct=float(100 * dd.mean() / ref['MAP']),
                          rel_se_pct=float(100 * dd.std(ddof=1) / np.sqrt(len(dd)) / ref['MAP']))), flush=True)
dev.close()
