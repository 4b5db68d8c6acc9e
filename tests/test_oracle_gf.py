"""CPU tests: the GF oracle against vectors produced by the reference itself
(scripts/make_golden.py ran gem/embedding/gf.py:91-101 under np.random.seed)."""
import numpy as np
import pytest

import oracle
from gem_amd.graph import edge_arrays
from conftest import golden_path


def _case(tag, graph):
    g = np.load(golden_path('gf_%s.npz' % tag))
    n, src, dst, w, _ = edge_arrays(graph)
    hp = dict(d=int(g['d']), eta=float(g['eta']), regu=float(g['regu']), max_iter=int(g['max_iter']))
    return g, n, src, dst, w, hp


@pytest.mark.parametrize('tag,gname', [('karate_ref_hp', 'karate'), ('karate_train', 'karate'), ('sbm1024_d32', 'sbm1024')])
def test_f64_oracle_reproduces_reference_python_loop(tag, gname, request):
    graph = request.getfixturevalue(gname)
    g, n, src, dst, w, hp = _case(tag, graph)
    # the init is numpy's legacy global RNG: reproducible from the seed (gf.py:92)
    np.random.seed(int(g['seed']))
    X0 = 0.01 * np.random.randn(n, hp['d'])
    assert np.array_equal(X0, g['X0'])
    X = oracle.gf_train_f64(n, src, dst, w, hp['d'], hp['eta'], hp['regu'], hp['max_iter'], X0)
    # same arithmetic, same order -> agreement to rounding of the fp64 dot (numpy uses pairwise/BLAS sums)
    np.testing.assert_allclose(X, g['X'], rtol=1e-9, atol=1e-13)


def test_python_loop_restatement_equals_the_vectors_gf_py_produced(karate):
    """oracle/gf_pyloop.py (bench.py's `cpu_baseline.python_loop`: gf.py:93-100 operation by operation) against the vectors produced by
    running gf.py itself -- same numpy ops in the same order, so equal to the last bit or two."""
    from oracle import gf_pyloop
    g, n, src, dst, w, hp = _case('karate_train', karate)
    X, visits, _ = gf_pyloop.gf_python_loop(src, dst, w, hp['eta'], hp['regu'], hp['max_iter'], g['X0'])
    assert visits == hp['max_iter'] * len(src)
    np.testing.assert_allclose(X, g['X'], rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize('tag,gname', [('karate_train', 'karate'), ('sbm1024_d32', 'sbm1024')])
def test_f32_oracle_tracks_f64(tag, gname, request):
    graph = request.getfixturevalue(gname)
    g, n, src, dst, w, hp = _case(tag, graph)
    X = oracle.gf_train_f32(n, src, dst, w, hp['d'], hp['eta'], hp['regu'], hp['max_iter'], g['X0'])
    scale = np.abs(g['X']).max()
    assert np.abs(X - g['X']).max() <= 2e-5 * max(scale, 1.0) + 1e-4 * scale


def test_objective_matches_numpy(karate):
    n, src, dst, w, _ = edge_arrays(karate)
    rng = np.random.RandomState(0)
    X = rng.randn(n, 6).astype(np.float32)
    f1, f2 = oracle.gf_objective(n, src, dst, w, 6, X)
    Xd = X.astype(np.float64)
    r = w - np.einsum('ij,ij->i', Xd[src], Xd[dst])
    assert np.isclose(f1, (r * r).sum(), rtol=1e-12)
    assert np.isclose(f2, (Xd * Xd).sum(), rtol=1e-12)


def test_reference_golden_gf_statistics(karate, sbm1024):
    """The reference's own GF goldens are only asserted through abs(mean(target - X)) < tol
    (tests/test_karate.py:78, tests/test_sbm.py:94).  The oracle, run with the reference
    hyper-parameters from a fresh init, satisfies the same assertion."""
    tgt = np.loadtxt(golden_path('ref_karate_GraphFactorization.txt'))
    n, src, dst, w, _ = edge_arrays(karate)
    X0 = 0.01 * np.random.RandomState(3).randn(n, 2)
    X = oracle.gf_train_f64(n, src, dst, w, 2, 1e-4, 1.0, 2000, X0)
    assert abs(np.mean(tgt - X)) < 0.3
    tgt = np.load(golden_path('ref_sbm_GraphFactorization.npz'))['X']
    n, src, dst, w, _ = edge_arrays(sbm1024)
    X0 = 0.01 * np.random.RandomState(4).randn(n, 128)
    X = oracle.gf_train_f32(n, src, dst, w, 128, 1e-4, 1.0, 20, X0)
    assert abs(np.mean(tgt - X)) < 1e-3
