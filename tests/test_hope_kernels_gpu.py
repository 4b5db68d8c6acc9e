"""GPU kernel-level parity: HOPE's SpMM, MFMA Gram and MFMA tall-skinny GEMM vs numpy (fp64 reference
of the same op; tolerance = fp32 accumulation)."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from gem_amd import _hip

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n,b,deg', [(34, 18, 3), (1000, 80, 10), (777, 1, 5), (300, 130, 90), (64, 512, 4)])
def test_spmm(n, b, deg):
    rng = np.random.RandomState(n + b)
    A = sp.random(n, n, density=min(1.0, deg / n), format='csr', random_state=rng, dtype=np.float64)
    A.data = rng.rand(len(A.data)) + 0.1
    X = rng.randn(n, b).astype(np.float32); W = rng.randn(n, b).astype(np.float32)
    rp = A.indptr.astype(np.int64); ci = A.indices.astype(np.int32); va = A.data.astype(np.float32)
    for wadd in (None, W):
        Y = np.empty((n, b), np.float32)
        _hip.check(_hip.lib().gemhip_hope_spmm(n, len(ci), _hip.ptr(rp, C.c_int64), _hip.ptr(ci, C.c_int32), _hip.ptr(va, C.c_float),
                                               0.37, b, _hip.ptr(X, C.c_float), _hip.ptr(wadd, C.c_float), _hip.ptr(Y, C.c_float)))
        ref = 0.37 * (sp.csr_matrix((va.astype(np.float64), ci, rp), shape=(n, n)) @ X.astype(np.float64)) + (0 if wadd is None else wadd)
        assert np.abs(Y - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1.0)


@pytest.mark.parametrize('n,m1,m2', [(34, 18, 18), (5000, 80, 80), (4097, 33, 70), (10000, 320, 96), (7, 3, 2)])
def test_gram_mfma(n, m1, m2):
    rng = np.random.RandomState(n)
    X = rng.randn(n, m1).astype(np.float32); Y = (rng.randn(n, m2) + np.arange(m2) * 0.01).astype(np.float32)   # asymmetric on purpose
    G = np.empty((m1, m2))
    _hip.check(_hip.lib().gemhip_hope_gram(n, m1, m2, _hip.ptr(X, C.c_float), _hip.ptr(Y, C.c_float), _hip.ptr(G, C.c_double)))
    ref = X.astype(np.float64).T @ Y.astype(np.float64)
    assert np.abs(G - ref).max() <= 3e-6 * np.sqrt(n) * 4 + 1e-5 * np.abs(ref).max()


@pytest.mark.parametrize('n,m,b2', [(34, 18, 16), (5000, 80, 80), (4097, 320, 64), (1000, 37, 5), (33, 9, 70), (3001, 448, 128), (515, 40, 160), (100, 65, 97)])
def test_tsgemm_mfma(n, m, b2):
    rng = np.random.RandomState(m)
    X = rng.randn(n, m).astype(np.float32); Cm = rng.randn(m, b2); S = rng.randn(n, b2).astype(np.float32)
    for src in (None, S):
        O = np.empty((n, b2), np.float32)
        _hip.check(_hip.lib().gemhip_hope_tsgemm(n, m, b2, _hip.ptr(X, C.c_float), _hip.ptr(Cm, C.c_double), -0.5,
                                                 _hip.ptr(src, C.c_float), _hip.ptr(O, C.c_float)))
        ref = (0 if src is None else src.astype(np.float64)) - 0.5 * (X.astype(np.float64) @ Cm.astype(np.float32).astype(np.float64))
        assert np.abs(O - ref).max() <= 1e-5 * np.sqrt(m) * max(np.abs(ref).max(), 1.0)
