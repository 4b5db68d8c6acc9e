"""Locally Linear Embedding on MI355X -- drop-in for gem.embedding.lle.LocallyLinearEmbedding (gem/embedding/lle.py:10-40).
SURVEY 8f row 3 ("next"): same SpMM + MFMA block-Krylov core as HOPE (gem_amd/csrc/hope.hip, gemhip_lle).

Reference (lle.py:23-35): graph.to_undirected(); A = adjacency, rows l1-normalised (sklearn normalize); u, s, vt =
svds(I - A, k=d+1, which='SM'); X = vt.T[:, 1:].  (Quirk not mirrored: when the graph carries no float weights the
reference's in-place `normalize(..., copy=False)` silently works on a copy and I - A is left UN-normalised.)
"""
import ctypes as C

import numpy as np

from gem_amd import _hip
from gem_amd.graph import to_csr
from gem_amd.embedding.lap import symmetric_arrays
from gem_amd.embedding.static_graph_embedding import StaticGraphEmbedding


class LocallyLinearEmbedding(StaticGraphEmbedding):
    hyper_params = {
        'method_name': 'lle_svd',
    }

    def __init__(self, *args, **kwargs):
        super(LocallyLinearEmbedding, self).__init__(*args, **kwargs)

    def learn_embedding(self, graph=None, edge_f=None, is_weighted=False, no_python=False, **_ignored):
        if not graph:
            raise ValueError('graph needed')
        n, src, dst, w = symmetric_arrays(graph)
        row_ptr, col, ww = to_csr(n, src, dst, w)
        k = int(self._d) + 1
        if k >= n:
            raise ValueError('LocallyLinearEmbedding needs d + 1 < n')
        _hip.require_device()
        V = np.empty((n, k), np.float32); sv = np.empty(k, np.float32)
        stats = (C.c_double * 12)()
        _hip.check(_hip.lib().gemhip_lle(n, len(col), _hip.ptr(row_ptr, C.c_int64), _hip.ptr(col, C.c_int32), _hip.ptr(ww, C.c_float), k,
                                         int(getattr(self, '_oversample', 16)), int(getattr(self, '_krylov_steps', 3)),
                                         int(getattr(self, '_max_restarts', 40)), float(getattr(self, '_tol', 1e-6)),
                                         int(getattr(self, '_seed', 20260923)), _hip.ptr(V, C.c_float), _hip.ptr(sv, C.c_float), stats))
        self._stats = dict(zip(('device_seconds', 'spmm_launches', 'spmm_columns', 'katz_terms', 'basis_columns', 'restarts', 'last_sigma_change', 'beta_sigma_max', 'host_eig_seconds', 'host_eig_calls', 'ritz_residual', 'spmm_seconds'), list(stats)))
        self._stats['solver'] = 'symmetric_chebyshev_filter' if self._stats['katz_terms'] < 0 else 'block_krylov'   # hope.hip: from 16384 nodes up
        _hip.warn_if_unconverged(self._stats, float(getattr(self, '_tol', 1e-6)), int(getattr(self, '_max_restarts', 40)), 'LocallyLinearEmbedding')
        self._singvals = sv.astype(np.float64)
        self._node_num = n
        self._X = V[:, 1:].astype(np.float64)                  # lle.py:33-34
        return self._X

    def _pair_matrix(self, X):
        sq = (X * X).sum(axis=1)
        return np.exp(-np.maximum(sq[:, None] + sq[None, :] - 2.0 * (X @ X.T), 0.0))

    def get_edge_weight(self, i, j):
        return np.exp(-np.power(np.linalg.norm(self._X[i, :] - self._X[j, :]), 2))
