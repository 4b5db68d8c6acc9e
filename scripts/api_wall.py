"""learn_embedding() API wall (graph ingest + H2D + kernels + D2H + float64 copy) vs kernel-only time at the BASELINE configs."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from gem_amd.embedding.gf import GraphFactorization
from gem_amd.embedding.hope import HOPE
from gem_amd.embedding.node2vec import node2vec
from gem_amd.graph import sbm_graph
g10k = sbm_graph(10000, 100000, 10, seed=20260924)
g100k = sbm_graph(100000, 1000000, 32, seed=20260925)
g1m = sbm_graph(1000000, 10000000, 100, seed=20260927)
def run(name, m, g, kern):
    t = time.time(); Y = m.learn_embedding(graph=g, edge_f=None, is_weighted=True, no_python=True); el = time.time() - t
    print('%-46s API wall %8.3f s   kernel-only %8.3f s   n=%d' % (name, el, kern(m), g.n), flush=True)
run('cfg2 GF d=128 max_iter=1000 SBM 10k/100k', GraphFactorization(d=128, eta=1e-4, regu=1.0, max_iter=1000), g10k, lambda m: m._stats['kernel_seconds'])
run('     GF d=128 max_iter=100  SBM 1M/10M', GraphFactorization(d=128, eta=1e-4, regu=1.0, max_iter=100), g1m, lambda m: m._stats['kernel_seconds'])
run('cfg3 HOPE d=128 beta=0.01   SBM 100k/1M', HOPE(d=128, beta=0.01), g100k, lambda m: m._stats['device_seconds'])
run('cfg4 node2vec d=128 r=10 l=80 k=10 SBM 1M/10M', node2vec(d=128, max_iter=1, walk_len=80, num_walks=10, con_size=10, ret_p=1, inout_p=1, seed=1), g1m,
    lambda m: m._stats['walk_seconds'] + m._stats['sgns_seconds'])
