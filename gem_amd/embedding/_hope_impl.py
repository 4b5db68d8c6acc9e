"""Host side of HOPE.learn_embedding: nx graph -> CSR in graph.nodes order -> libgem_hip.so."""
import ctypes as C

import numpy as np

from gem_amd import _hip
from gem_amd.graph import edge_arrays, to_csr


def learn(model, graph):
    import time
    t0 = time.perf_counter()
    n, src, dst, w, order = edge_arrays(graph)
    if order is not None:
        # hope.py:28 builds A with nx.to_numpy_matrix(graph): row/column index = position in graph.nodes,
        # NOT the node id.  Row r of the returned X belongs to the r-th inserted node.
        pos = np.empty(n, dtype=np.int64)
        pos[order] = np.arange(n)
        src, dst = pos[src].astype(np.int32), pos[dst].astype(np.int32)
    row_ptr, col, ww = to_csr(n, src, dst, w)
    t1 = time.perf_counter()
    d = int(model._d)
    k = d // 2
    if k < 1 or k >= n:
        raise ValueError('HOPE needs 1 <= d//2 < n (scipy svds: k must satisfy 0 < k < min(shape))')
    _hip.require_device()
    U = np.empty((n, k), dtype=np.float32); V = np.empty((n, k), dtype=np.float32); sig = np.empty(k, dtype=np.float32)
    stats = (C.c_double * 12)()
    _hip.check(_hip.lib().gemhip_hope(n, len(col), _hip.ptr(row_ptr, C.c_int64), _hip.ptr(col, C.c_int32), _hip.ptr(ww, C.c_float),
                                      float(model._beta), k, int(getattr(model, '_oversample', 16)),
                                      int(getattr(model, '_krylov_steps', 3)), int(getattr(model, '_max_restarts', 20)),
                                      float(getattr(model, '_tol', 1e-5)), int(getattr(model, '_seed', 20260923)),
                                      _hip.ptr(U, C.c_float), _hip.ptr(V, C.c_float), _hip.ptr(sig, C.c_float), stats))
    t2 = time.perf_counter()
    model._sigma = sig.astype(np.float64)
    model._stats = dict(zip(('device_seconds', 'spmm_launches', 'spmm_columns', 'katz_terms', 'basis_columns', 'restarts',
                             'last_sigma_change', 'beta_sigma_max', 'host_eig_seconds', 'host_eig_calls', 'ritz_residual', 'spmm_seconds'), list(stats)))
    model._stats['solver'] = 'symmetric_chebyshev_filter' if model._stats['katz_terms'] == 0 else 'block_krylov'   # hope.hip: A == A^T takes the eigen-path
    _hip.warn_if_unconverged(model._stats, float(getattr(model, '_tol', 1e-5)), int(getattr(model, '_max_restarts', 20)), 'HOPE')
    model._node_num = n
    if getattr(model, '_verbose', False):
        # hope.py:38-40 prints ||u diag(s) vt - S||_F of the dense S it formed; here ||S (I - V V^T)||_F^2 (32 deflated probe columns pushed through the Katz
        # series) + ||S V - U Sigma||_F^2 (exact; ~0 for a converged solve, so a wrong U shows as it would in the reference's print) -- gemhip_hope_svd_error_uv.
        # Opt-in (HOPE(..., verbose=True)): it rebuilds the plan outside the timed solve
        err = C.c_double(); fro2 = C.c_double(); uside = C.c_double()
        _hip.check(_hip.lib().gemhip_hope_svd_error_uv(n, len(col), _hip.ptr(row_ptr, C.c_int64), _hip.ptr(col, C.c_int32), _hip.ptr(ww, C.c_float),
                                                       float(model._beta), k, _hip.ptr(sig, C.c_float), _hip.ptr(U, C.c_float), _hip.ptr(V, C.c_float), 32,
                                                       int(getattr(model, '_seed', 20260923)), C.byref(err), C.byref(fro2), C.byref(uside)))
        model._svd_error_u_side = uside.value
        model._svd_error = err.value
        print('SVD error (low rank): %f' % err.value)
    X64 = np.empty((n, 2 * k), dtype=np.float64)              # [U sqrt(S) | V sqrt(S)] (hope.py:34-36), each half converted in place: one pass, no float32 concatenate
    X64[:, :k] = U
    X64[:, k:] = V
    model._api_wall = _hip.api_wall(t0, t1, t2, time.perf_counter())
    return X64
