"""Extra data points for DESIGN.md: HOPE beyond cfg3, Laplacian Eigenmaps / LLE at cfg3 size (API wall and device time)."""
import sys, time
sys.path.insert(0, '.')
from gem_amd.embedding.hope import HOPE
from gem_amd.embedding.lap import LaplacianEigenmaps
from gem_amd.embedding.lle import LocallyLinearEmbedding
from gem_amd.graph import sbm_graph


def run(name, m, g):
    for rep in range(2):
        t = time.time(); m.learn_embedding(graph=g, edge_f=None, is_weighted=True, no_python=True); el = time.time() - t
    st = m._stats
    print('%-40s n=%-8d API wall %.3f s  stats %s' % (name, g.n, el, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()}), flush=True)


g100k = sbm_graph(100000, 1000000, 32, seed=20260925)
g1m = sbm_graph(1000000, 10000000, 100, seed=20260927)
run('HOPE d=128 beta=0.01 SBM 100k/1M', HOPE(d=128, beta=0.01), g100k)
run('HOPE d=128 beta=0.01 SBM 1M/10M', HOPE(d=128, beta=0.01), g1m)
run('LaplacianEigenmaps d=64 SBM 100k/1M', LaplacianEigenmaps(d=64), g100k)
run('LocallyLinearEmbedding d=64 SBM 100k/1M', LocallyLinearEmbedding(d=64), g100k)
