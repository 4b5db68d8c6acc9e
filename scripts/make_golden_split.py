#!/usr/bin/env python3
"""tests/golden/split_ref.npz: train/test edge sets produced by the REFERENCE split_di_graph_to_train_test
(gem/utils/evaluation_util.py:39-53) under np.random.seed(17): karate as a directed graph, SBM-1024 as undirected."""
import os, sys
sys.path.insert(0, '/root/reference')
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(1, os.path.join(ROOT, 'tests')); sys.path.insert(1, ROOT)
from gem.utils import evaluation_util                     # the reference's
import importlib.util
spec = importlib.util.spec_from_file_location('conftest_local', os.path.join(ROOT, 'tests', 'conftest.py'))
# build the graphs without importing our `gem` alias
import networkx as nx
def load_karate():
    G = nx.DiGraph()
    for line in open(os.path.join(ROOT, 'tests/golden/karate.edgelist')):
        t = line.split(); G.add_edge(int(t[0]), int(t[1]), weight=float(t[2]) if len(t) == 3 else 1.0)
    return G.to_directed()
def load_sbm():
    e = np.load(os.path.join(ROOT, 'tests/golden/sbm1024_edges.npy')); nodes = np.load(os.path.join(ROOT, 'tests/golden/sbm1024_nodes.npy'))
    G = nx.DiGraph(); G.add_nodes_from(nodes.tolist()); G.add_edges_from(map(tuple, e.tolist())); return G
out = {}
for name, G, und in (('karate', load_karate(), False), ('sbm', load_sbm(), True)):
    np.random.seed(17)
    tr, te = evaluation_util.split_di_graph_to_train_test(G, 0.8, is_undirected=und)
    n = len(G.nodes)
    for tag, g in (('train', tr), ('test', te)):
        out['%s_%s' % (name, tag)] = np.unique(np.array([a * n + b for a, b in g.edges()], dtype=np.int64))
np.savez_compressed(os.path.join(ROOT, 'tests/golden/split_ref.npz'), **out)
print({k: len(v) for k, v in out.items()})
