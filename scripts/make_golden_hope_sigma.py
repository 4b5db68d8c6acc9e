#!/usr/bin/env python3
"""HOPE at BASELINE configs[2] (SBM 100k nodes / 1M edges, d=128 -> k=64, beta=0.01): the 64 singular values hope.py:33 would
return (`svds(S, k=d//2)`), computed on the CPU with scipy's ARPACK svds on the implicit Katz-series operator
(oracle/hope_oracle.py hope_operator_series; the literal dense S does not fit in memory at this size), tol=1e-9.
Writes tests/golden/hope_sigma_sbm100k.json (graph = bench.py's: gem_amd.graph.sbm_graph(100000, 1000000, 32, 20260923+4))."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
from gem_amd.graph import sbm_graph, orient_randomly
from oracle import hope_oracle
DIRECTED = '--directed' in sys.argv      # the same SBM with every undirected edge kept in one random direction (orient_randomly, seed 1): A != A^T
P = dict(n=100000, edges=1000000, blocks=32, seed=20260923 + 4, d=128, beta=0.01)
g = sbm_graph(P['n'], P['edges'], P['blocks'], P['seed'])
if DIRECTED:
    g = orient_randomly(g, 1); P['orient_seed'] = 1
A = sp.csr_matrix((np.ones(g.number_of_edges()), (g.src, g.dst)), shape=(g.n, g.n))
t = time.time()
_, s = hope_oracle.hope_operator_series(A, P['beta'], P['d'], tol=1e-9)
el = time.time() - t
json.dump({'params': P, 'engine': 'scipy.sparse.linalg.svds (ARPACK) on the Katz-series operator, tol=1e-9', 'seconds': el,
           'sigma_ascending': [float(v) for v in s]}, open(os.path.join(ROOT, 'tests', 'golden', 'hope_sigma_sbm100k%s.json' % ('_directed' if DIRECTED else '')), 'w'), indent=1)
print('done in %.0fs' % el, s[:3], s[-3:])
