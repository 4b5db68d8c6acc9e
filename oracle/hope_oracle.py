"""oracle/hope_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of HOPE.learn_embedding (gem/embedding/hope.py:23-41).  Pinned by
  * tests/golden/ref_karate_HOPE.txt      the reference's own golden (tests/karate_res/HOPE.txt,
                                          asserted with np.allclose in tests/test_karate.py:42-45,76)
  * tests/golden/hope_{karate_d4,sbm1024_d32}.npz  produced by scripts/make_golden.py RUNNING hope.py
  * tests/golden/hope_sbm1024_sigma.npy   exact singular values of S
up to the sign of each singular-vector pair (ARPACK's signs are arbitrary).

hope_dense     follows hope.py line by line with a full LAPACK SVD in place of ARPACK (same triplets).
hope_operator  the same SVD through scipy svds on a LinearOperator that applies S and S^T with sparse
               LU solves -- never forms S; the CPU baseline at sizes where the literal path cannot run
               (n = 100k needs three 80 GB matrices).
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as sla


def adjacency(n, src, dst, w=None, order=None):
    """CSR of A indexed in graph.nodes order (hope.py:28: nx.to_numpy_matrix(graph))."""
    src = np.asarray(src); dst = np.asarray(dst)
    if order is not None:
        pos = np.empty(n, dtype=np.int64); pos[np.asarray(order)] = np.arange(n)
        src, dst = pos[src], pos[dst]
    vals = np.ones(len(src)) if w is None else np.asarray(w, dtype=np.float64)
    return sp.csr_matrix((vals, (src, dst)), shape=(n, n))


def hope_dense(A, beta, d):
    """hope.py:28-36 literally (dense), LAPACK SVD; returns (X, sigma_ascending)."""
    A = np.asarray(A.todense()) if sp.issparse(A) else np.asarray(A)
    n = A.shape[0]
    m_g = np.eye(n) - beta * A                      # :29
    m_l = beta * A                                  # :30
    S = np.dot(np.linalg.inv(m_g), m_l)             # :31
    k = d // 2
    u, s, vt = np.linalg.svd(S)
    u, s, vt = u[:, :k][:, ::-1], s[:k][::-1], vt[:k][::-1]       # svds order: ascending (:33)
    X1 = u * np.sqrt(s)                             # :34
    X2 = vt.T * np.sqrt(s)                          # :35
    return np.concatenate((X1, X2), axis=1), s      # :36


def hope_operator(A, beta, d, tol=0, ncv=None):
    """svds(LinearOperator(S, S^T), k=d//2) with S x = (I - beta A)^-1 (beta A x) via sparse LU."""
    A = sp.csc_matrix(A, dtype=np.float64)
    n = A.shape[0]
    M = sp.identity(n, format='csc') - beta * A
    lu = sla.splu(M)
    lut = sla.splu(M.T.tocsc())
    At = A.T.tocsr(); Ar = A.tocsr()
    op = sla.LinearOperator((n, n), matvec=lambda x: lu.solve(beta * (Ar @ x)), rmatvec=lambda y: beta * (At @ lut.solve(y)),
                            dtype=np.float64)
    k = d // 2
    u, s, vt = sla.svds(op, k=k, tol=tol, ncv=ncv)
    order = np.argsort(s)
    u, s, vt = u[:, order], s[order], vt[order]
    return np.concatenate((u * np.sqrt(s), vt.T * np.sqrt(s)), axis=1), s


def hope_operator_series(A, beta, d, tol=0, ncv=None, series_tol=1e-15):
    """Same SVD, S applied through its Katz series S x = sum_{t>=1} (beta A)^t x (converges when
    beta*rho(A) < 1), i.e. sparse mat-vecs only: the practical CPU baseline at n >= 1e4, where the sparse LU
    of (I - beta A) fills in almost completely for these random graphs."""
    A = sp.csr_matrix(A, dtype=np.float64)
    n = A.shape[0]
    At = A.T.tocsr()
    x = np.ones(n) / np.sqrt(n)
    for _ in range(50):
        x = At @ (A @ x); nx = np.linalg.norm(x); x /= nx
    br = beta * np.sqrt(nx) * 1.05
    if not br < 1:
        raise ValueError('Katz series diverges: beta*sigma_max(A) = %.3f' % br)
    terms = int(np.ceil(np.log(series_tol) / np.log(br)))

    def mv(x):
        w = beta * (A @ x); z = w
        for _ in range(terms):
            z = w + beta * (A @ z)
        return z

    def rmv(y):
        r = y
        for _ in range(terms):
            r = y + beta * (At @ r)
        return beta * (At @ r)
    op = sla.LinearOperator((n, n), matvec=mv, rmatvec=rmv, dtype=np.float64)
    k = d // 2
    u, s, vt = sla.svds(op, k=k, tol=tol, ncv=ncv)
    order = np.argsort(s)
    u, s, vt = u[:, order], s[order], vt[order]
    return np.concatenate((u * np.sqrt(s), vt.T * np.sqrt(s)), axis=1), s


def align_signs(X, Xref, d):
    """Flip singular-vector pairs of X to the signs of Xref (pair j = columns j and k+j)."""
    k = d // 2
    X = X.copy()
    for j in range(k):
        if np.dot(X[:, j], Xref[:, j]) + np.dot(X[:, k + j], Xref[:, k + j]) < 0:
            X[:, j] *= -1; X[:, k + j] *= -1
    return X


def lap_eigmap_dense(n, src, dst, w, d):
    """gem/embedding/lap.py:21-37 with a dense symmetric eigensolver: the d+1 smallest eigenpairs of
    L_sym = I - D^-1/2 A D^-1/2 (nx.normalized_laplacian_matrix: isolated nodes get D^-1/2 = 0), X = v[:, 1:].
    src/dst/w: the SYMMETRIC adjacency (both directions listed)."""
    A = np.zeros((n, n))
    A[np.asarray(src), np.asarray(dst)] = np.asarray(w, dtype=np.float64)
    deg = A.sum(axis=1)
    with np.errstate(divide='ignore'):
        dinv = np.where(deg > 0, 1.0 / np.sqrt(deg), 0.0)
    L = np.eye(n) - dinv[:, None] * A * dinv[None, :]
    wv, v = np.linalg.eigh(L)
    return v[:, 1:d + 1], wv[:d + 1]


def lle_dense(n, src, dst, w, d):
    """gem/embedding/lle.py:23-35 with a dense SVD: rows of the symmetric adjacency l1-normalised, the d+1 smallest
    singular triplets of I - P, X = vt.T[:, 1:] (ascending singular value).  Returns (X, smallest singular values)."""
    A = np.zeros((n, n))
    A[np.asarray(src), np.asarray(dst)] = np.asarray(w, dtype=np.float64)
    l1 = np.abs(A).sum(axis=1, keepdims=True)
    P = np.divide(A, l1, out=np.zeros_like(A), where=l1 > 0)
    u, s, vt = np.linalg.svd(np.eye(n) - P)
    V = vt[::-1].T                                    # ascending singular value
    return V[:, 1:d + 1], s[::-1][:d + 1]
