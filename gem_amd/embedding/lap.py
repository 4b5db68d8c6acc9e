"""Laplacian Eigenmaps on MI355X -- drop-in for gem.embedding.lap.LaplacianEigenmaps (gem/embedding/lap.py:8-42).
SURVEY 8f row 3 ("next"): the smallest eigenvectors of the normalised Laplacian reuse HOPE's SpMM + MFMA block-Krylov
machinery (gem_amd/csrc/hope.hip, gemhip_lap_eigmap).

Reference (lap.py:21-37): graph.to_undirected(); L = nx.normalized_laplacian_matrix(graph); w, v = eigs(L, k=d+1,
which='SM'); sort by w; X = v[:, 1:] (real part).  Rows in graph.nodes order.  Eigenvector signs are arbitrary.
"""
import ctypes as C

import numpy as np

from gem_amd import _hip
from gem_amd.graph import EdgeListGraph, to_csr
from gem_amd.embedding.static_graph_embedding import StaticGraphEmbedding


def symmetric_arrays(graph):
    """(n, src, dst, w) of graph.to_undirected() with both directions listed, indexed by position in graph.nodes."""
    if isinstance(graph, EdgeListGraph):
        n = graph.n
        s, d = graph.src.astype(np.int64), graph.dst.astype(np.int64)
        w = np.ones(len(s), np.float32) if graph.w is None else graph.w
        lo, hi = np.minimum(s, d), np.maximum(s, d)
        key, first = np.unique(lo * n + hi, return_index=True)
        lo, hi, w = key // n, key % n, w[first]
    else:
        und = graph.to_undirected()
        n = len(und.nodes)
        pos = {v: r for r, v in enumerate(und.nodes)}
        e = [(pos[a], pos[b], wt) for a, b, wt in und.edges(data='weight', default=1)]
        lo = np.array([min(a, b) for a, b, _ in e], np.int64); hi = np.array([max(a, b) for a, b, _ in e], np.int64)
        w = np.array([wt for _, _, wt in e], np.float32)
    loop = lo == hi
    src = np.concatenate([lo, hi[~loop]]); dst = np.concatenate([hi, lo[~loop]]); ww = np.concatenate([w, w[~loop]])
    return n, src.astype(np.int32), dst.astype(np.int32), ww.astype(np.float32)


class LaplacianEigenmaps(StaticGraphEmbedding):
    hyper_params = {
        'method_name': 'lap_eigmap_svd',
    }

    def __init__(self, *args, **kwargs):
        super(LaplacianEigenmaps, self).__init__(*args, **kwargs)

    def learn_embedding(self, graph=None, edge_f=None, is_weighted=False, no_python=False, **_ignored):
        if not graph:
            raise ValueError('graph needed')
        n, src, dst, w = symmetric_arrays(graph)
        row_ptr, col, ww = to_csr(n, src, dst, w)
        d = int(self._d)
        k = d + 1
        if k >= n:
            raise ValueError('LaplacianEigenmaps needs d + 1 < n')
        _hip.require_device()
        V = np.empty((n, k), np.float32); ev = np.empty(k, np.float32)
        stats = (C.c_double * 12)()
        _hip.check(_hip.lib().gemhip_lap_eigmap(n, len(col), _hip.ptr(row_ptr, C.c_int64), _hip.ptr(col, C.c_int32), _hip.ptr(ww, C.c_float),
                                                k, int(getattr(self, '_oversample', 16)), int(getattr(self, '_krylov_steps', 3)),
                                                int(getattr(self, '_max_restarts', 30)), float(getattr(self, '_tol', 1e-6)),
                                                int(getattr(self, '_seed', 20260923)), _hip.ptr(V, C.c_float), _hip.ptr(ev, C.c_float), stats))
        self._stats = dict(zip(('device_seconds', 'spmm_launches', 'spmm_columns', 'katz_terms', 'basis_columns', 'restarts', 'last_sigma_change', 'beta_sigma_max', 'host_eig_seconds', 'host_eig_calls', 'ritz_residual', 'spmm_seconds'), list(stats)))
        self._stats['solver'] = 'symmetric_chebyshev_filter' if self._stats['katz_terms'] < 0 else 'block_krylov'   # hope.hip: from 16384 nodes up
        _hip.warn_if_unconverged(self._stats, float(getattr(self, '_tol', 1e-6)), int(getattr(self, '_max_restarts', 30)), 'LaplacianEigenmaps')
        self._eigvals = ev.astype(np.float64)
        self._node_num = n
        self._X = V[:, 1:].astype(np.float64)                  # lap.py:32: drop the trivial eigenvector
        return self._X

    def _pair_matrix(self, X):
        sq = (X * X).sum(axis=1)
        return np.exp(-np.maximum(sq[:, None] + sq[None, :] - 2.0 * (X @ X.T), 0.0))

    def get_edge_weight(self, i, j):
        return np.exp(-np.power(np.linalg.norm(self._X[i, :] - self._X[j, :]), 2))
