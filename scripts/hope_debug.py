import sys; sys.path.insert(0,'/root/repo')
from gem_amd.embedding.hope import HOPE
from gem_amd.graph import sbm_graph
g = sbm_graph(100000, 1000000, 32, seed=20260927)
m = HOPE(d=128, beta=0.01, tol=1e-7, max_restarts=40); m.learn_embedding(graph=g)
