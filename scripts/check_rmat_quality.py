"""node2vec on a power-law (R-MAT scale 13) graph: Hogwild GPU path vs the sequential CPU oracle (MAP)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import oracle
from gem_amd.graph import rmat_graph, edge_arrays
from gem_amd.embedding.node2vec import node2vec
from gem_amd.evaluation import reconstruction as gr
g = rmat_graph(13, 160000, seed=20260928)
n, src, dst, w, _ = edge_arrays(g)
d = 32
m = node2vec(d=d, max_iter=1, walk_len=80, num_walks=10, con_size=10, ret_p=1, inout_p=1, seed=1)
for seed in (1, 2):
    node2vec.hyper_params['seed'] = seed
    m = node2vec(d=d, max_iter=1, walk_len=80, num_walks=10, con_size=10, ret_p=1, inout_p=1, seed=seed)
    t = time.time(); Y = m.learn_embedding(graph=g); el = time.time() - t
    print('gpu seed', seed, 'MAP %.4f' % gr.evaluateStaticGraphReconstruction(g, m, Y, None)[0], '%.2fs' % el, flush=True)
t = time.time(); X, _ = oracle.n2v_train(n, src, dst, None, d, 80, 10, 10, 1, 1.0, 1.0, 1, 11); el = time.time() - t
print('oracle sequential MAP %.4f' % gr.evaluateStaticGraphReconstruction(g, m, X.astype(np.float64), None)[0], '%.1fs' % el, flush=True)
