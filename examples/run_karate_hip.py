#!/usr/bin/env python3
"""The flow of GEM's examples/run_karate.py on the MI355X backend, through GEM's own import paths (the `gem` alias
package): load the karate edge list, fit every in-scope method, report training time and graph-reconstruction MAP.

    MPLBACKEND=Agg python examples/run_karate_hip.py [-node2vec 1]
"""
import os
import sys
from argparse import ArgumentParser
from time import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

from gem.utils import graph_util
from gem.evaluation import evaluate_graph_reconstruction as gr
from gem.embedding.gf import GraphFactorization
from gem.embedding.hope import HOPE
from gem.embedding.lap import LaplacianEigenmaps
from gem.embedding.lle import LocallyLinearEmbedding
from gem.embedding.node2vec import node2vec

if __name__ == '__main__':
    parser = ArgumentParser(description='Graph embedding on the karate graph, MI355X backend')
    parser.add_argument('-node2vec', '--node2vec', default='1')
    parser.add_argument('-sdne', '--sdne', default='0', help="1: end with GEM's sixth model, SDNE (a Keras auto-encoder, out of scope here: the stub constructs and refuses to train)")
    args = parser.parse_args()
    run_n2v = bool(int(args.node2vec))
    here = os.path.dirname(os.path.abspath(__file__))
    G = graph_util.loadGraphFromEdgeListTxt(os.path.join(here, '..', 'tests', 'golden', 'karate.edgelist'), directed=True).to_directed()
    models = [GraphFactorization(d=2, max_iter=50000, eta=1 * 10 ** -4, regu=1.0, data_set='karate'), HOPE(d=4, beta=0.01),
              LaplacianEigenmaps(d=2), LocallyLinearEmbedding(d=2)]
    if run_n2v:
        models.append(node2vec(d=2, max_iter=1, walk_len=80, num_walks=10, con_size=10, ret_p=1, inout_p=1))
    if bool(int(args.sdne)):
        from gem.embedding.sdne import SDNE
        models.append(SDNE(d=2, beta=5, alpha=1e-5, nu1=1e-6, nu2=1e-6, K=3, n_units=[50, 15], n_iter=50, xeta=0.01, n_batch=500))
    for embedding in models:
        print('Num nodes: %d, num edges: %d' % (G.number_of_nodes(), G.number_of_edges()))
        t1 = time()
        Y = embedding.learn_embedding(graph=G, edge_f=None, is_weighted=True, no_python=True)
        print(embedding.get_method_name() + ':\n\tTraining time: %f' % (time() - t1))
        MAP, prec_curv, err, err_baseline = gr.evaluateStaticGraphReconstruction(G, embedding, Y, None)
        print(("\tMAP: {} \t preccision curve: {}\n\n" + '-' * 100).format(MAP, prec_curv[:5]))
