"""CPU: the numpy mirror of hope.hip's symmetric eigen-path (tests/hope_sym_mirror.py) against scipy -- the algorithm's logic, without
a GPU: selection of the k largest |f(lambda)| from BOTH ends of the spectrum, u = sign(f(lambda)) q, convergence within the cycle budget,
and the SpMM-column count that motivates the path (the block-Krylov solver on S^T S applies 18 SpMMs per Krylov step)."""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as sla

from gem_amd.graph import sbm_graph
from oracle import hope_oracle
from hope_sym_mirror import sym_filter_svd


def test_sbm_matches_the_dense_literal_hope():
    """SBM 1024 nodes: S = inv(I - beta A) beta A formed densely and decomposed like hope.py:28-33 (oracle/hope_oracle.py)."""
    g = sbm_graph(1024, 10240, 4, seed=3)
    A = sp.csr_matrix((np.ones(g.number_of_edges(), np.float32), (g.src, g.dst)), shape=(g.n, g.n))
    assert abs(A - A.T).nnz == 0
    trace = []
    s, U, V, info = sym_filter_svd(A, 0.01, 8, trace=trace)
    assert info['converged'] and info['cycles'] <= 20, trace
    Xo, so = hope_oracle.hope_dense(np.asarray(A.todense(), dtype=np.float64), 0.01, 16)      # sigma ascending
    assert np.allclose(s[::-1], so, rtol=5e-5), np.abs(s[::-1] / so - 1).max()
    # the 4 community triplets are separated: U sqrt(s), V sqrt(s) agree with hope.py's output up to the sign of each pair
    X = np.concatenate([(U * np.sqrt(s))[:, ::-1], (V * np.sqrt(s))[:, ::-1]], axis=1)          # ascending sigma like svds
    for j in (7, 6, 5, 4):
        for half in (0, 8):
            c = np.dot(X[:, half + j], Xo[:, half + j]) / (np.linalg.norm(X[:, half + j]) * np.linalg.norm(Xo[:, half + j]))
            assert abs(c) > 1 - 1e-6
    assert np.abs(U.T @ U - np.eye(8)).max() < 1e-5


def test_two_sided_spectrum_and_negative_eigenvalues():
    """Weighted bipartite graph: spectrum +-lambda.  The k largest |f| mix both ends and a negative eigenvalue has u = -v."""
    n = 4096
    rs = np.random.RandomState(5)
    a = rs.randint(0, n // 2, 30000); b = rs.randint(n // 2, n, 30000)
    key = np.unique(a.astype(np.int64) * n + b); a = (key // n); b = (key % n)
    w = (rs.rand(len(a)) + 0.5).astype(np.float32)
    A = sp.csr_matrix((np.concatenate([w, w]), (np.concatenate([a, b]), np.concatenate([b, a]))), shape=(n, n))
    beta, k = 0.01, 12
    s, U, V, info = sym_filter_svd(A, beta, k)
    assert info['converged']
    lam = np.concatenate([sla.eigsh(A.astype(np.float64), k=2 * k, which='LA')[0], sla.eigsh(A.astype(np.float64), k=4, which='SA')[0]])
    ref = np.sort(np.abs(beta * lam / (1 - beta * lam)))[::-1][:k]
    assert np.allclose(s, ref, rtol=5e-5), np.abs(s / ref - 1).max()
    assert np.allclose(U[:, 0], V[:, 0]) and np.allclose(U[:, 1], -V[:, 1])          # Perron pair, then its mirror -lambda_max
    assert abs(s[1] - beta * lam.max() / (1 + beta * lam.max())) < 1e-5 * s[1]       # |f(-lambda_max)|
    # every returned triplet on its own: S v = sigma u
    Sv = V.copy(); Wk = beta * (A @ V); Sv = Wk.copy()
    for _ in range(60):
        Sv = Wk + beta * (A @ Sv)
    assert np.linalg.norm(Sv - U * s, axis=0).max() < 1e-2 * s[0]


def test_work_against_the_block_krylov_path():
    """SBM 8192/81920, k=16: the filter needs a few thousand SpMM columns; one block-Krylov cycle on S^T S (3 steps x 18 SpMMs of 32
    columns + 9 for the last block) is ~2000 columns and the GPU solver needs ~8 cycles."""
    g = sbm_graph(8192, 81920, 8, seed=11)
    A = sp.csr_matrix((np.ones(g.number_of_edges(), np.float32), (g.src, g.dst)), shape=(g.n, g.n))
    s, U, V, info = sym_filter_svd(A, 0.01, 16)
    assert info['converged'] and info['cycles'] <= 12 and info['columns'] < 4000, info
