"""HOPE on MI355X -- drop-in for gem.embedding.hope.HOPE (gem/embedding/hope.py:8-44).

Reference algorithm (hope.py:28-36): S = (I - beta A)^-1 (beta A) (Katz proximity),
u, s, vt = svds(S, k=d//2); X = [u sqrt(s) | vt.T sqrt(s)] with singular values in
svds' ASCENDING order and rows in graph.nodes insertion order.

The device path never forms S: see gem_amd/csrc/hope.hip.
"""
import numpy as np

from gem_amd.embedding.static_graph_embedding import StaticGraphEmbedding


class HOPE(StaticGraphEmbedding):
    hyper_params = {
        'method_name': 'hope_gsvd',
    }

    def __init__(self, *args, **kwargs):
        super(HOPE, self).__init__(*args, **kwargs)

    def learn_embedding(self, graph=None, edge_f=None, is_weighted=False, no_python=False, **_ignored):
        if not graph:
            raise ValueError('graph needed')
        if getattr(self, '_n_gpus', 1) not in (1, None):
            # HOPE is "replicas only" -- cfg3 fits one GPU and north_star shards GF and node2vec; there is no N-GPU HOPE to fall back from silently
            raise ValueError('HOPE does not shard (n_gpus=%r): run one replica per GPU' % (self._n_gpus,))
        from gem_amd.embedding import _hope_impl
        self._X = _hope_impl.learn(self, graph)
        return self._X

    def _pair_matrix(self, X):
        k = self._d // 2
        return X[:, :k] @ X[:, k:].T

    def get_edge_weight(self, i, j):
        k = self._d // 2
        return np.dot(self._X[i, :k], self._X[j, k:])
