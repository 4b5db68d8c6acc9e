# ARCHIVED (round 3): drives the shared-negatives kernel (flag 64), which was removed from the library; kept because committed profiles were produced with it
"""Opt-in shared-negatives SGNS: MAP at the 16k-node reference point and time at SBM 1M/10M."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from gem_amd import _hip, multi_gpu
from gem_amd.graph import sbm_graph, edge_arrays, to_csr
from gem_amd.embedding.node2vec import node2vec
from gem_amd.evaluation import reconstruction as gr
ref = json.load(open(os.path.join(ROOT, 'tests/golden/n2v_ref_16k.json'))); p = ref['params']
g = sbm_graph(p['n'], p['edges'], p['blocks'], p['seed'])
for flags in (11, 11 | 64):
    m = node2vec(d=128, max_iter=1, walk_len=80, num_walks=10, con_size=10, ret_p=1, inout_p=1, seed=3, flags=flags)
    Y = m.learn_embedding(graph=g); node2vec.hyper_params.pop('flags', None)
    print('16k flags', flags, 'MAP %.4f' % gr.evaluateStaticGraphReconstruction(g, m, Y, None)[0], '(SNAP race-free %.4f)' % ref['snap']['t1']['MAP'], flush=True)
g = sbm_graph(1000000, 10000000, 100, seed=20260927)
n, src, dst, w, _ = edge_arrays(g); row_ptr, col, ww = to_csr(n, src, dst, w)
b = multi_gpu.HipBackendN2V(n, row_ptr, col, ww, 128)
m = b.num_start_nodes(); b.walks(1.0, 1.0, 10, 80, 1, 11, 0, m * 10); b.vocab(); b.build_unigram()
rng = np.random.RandomState(0); nodes = rng.choice(n, 256, replace=False)
for flags in (11 | 64, 11):
    b.init_tables(1); torch.cuda.synchronize(); t = time.time()
    b.train(10, 1, 0, 0, m * 10, m * 800, 0, 1, flags); torch.cuda.synchronize(); el = time.time() - t
    ap = gr.sampled_ap_gpu(g, None, b.P.cpu().numpy(), nodes)
    print('1M flags', flags, 'sgns %.2f s' % el, 'sampled MAP %.4f' % ap.mean(), flush=True)
