"""CPU tests: the vectorised MAP / precision curve against values computed by the
reference's own evaluator (tests/golden/map_ref.json, scripts/make_golden.py)."""
import json

import numpy as np
import pytest

from gem_amd.embedding.gf import GraphFactorization
from gem_amd.embedding.hope import HOPE
from gem_amd.embedding.node2vec import node2vec
from gem_amd.evaluation import reconstruction as gr
from conftest import golden_path

with open(golden_path('map_ref.json')) as fh:
    REF = json.load(fh)


def test_map_matches_reference_on_gf_goldens(karate, sbm1024):
    m = GraphFactorization(d=2, max_iter=1, eta=1e-4, regu=1.0)
    X = np.loadtxt(golden_path('ref_karate_GraphFactorization.txt'))
    assert gr.evaluateStaticGraphReconstruction(karate, m, X, None)[0] == pytest.approx(REF['karate_gf_golden'], abs=1e-12)
    m = GraphFactorization(d=128, max_iter=1, eta=1e-4, regu=1.0)
    X = np.load(golden_path('ref_sbm_GraphFactorization.npz'))['X'].astype(np.float64)
    # the golden was saved as float32: MAP is rank based, allow the few near-tie swaps that may cause
    assert gr.evaluateStaticGraphReconstruction(sbm1024, m, X, None)[0] == pytest.approx(REF['sbm1024_gf_golden'], abs=2e-4)
    g = np.load(golden_path('gf_karate_train.npz'))
    m = GraphFactorization(d=8, max_iter=1, eta=0.05, regu=0.01)
    assert gr.evaluateStaticGraphReconstruction(karate, m, g['X'], None)[0] == pytest.approx(REF['karate_gf_train'], abs=1e-12)
    g = np.load(golden_path('gf_sbm1024_d32.npz'))
    m = GraphFactorization(d=32, max_iter=1, eta=0.02, regu=0.01)
    assert gr.evaluateStaticGraphReconstruction(sbm1024, m, g['X'], None)[0] == pytest.approx(
        REF['sbm1024_gf_d32_5sweeps'], abs=1e-12)


def test_map_matches_reference_on_hope_and_n2v_goldens(karate, sbm1024):
    m = HOPE(d=4, beta=0.01)
    X = np.loadtxt(golden_path('ref_karate_HOPE.txt'))
    assert gr.evaluateStaticGraphReconstruction(karate, m, X, None)[0] == pytest.approx(REF['karate_hope_golden'], abs=1e-12)
    X = np.load(golden_path('hope_karate_d4.npz'))['X']
    assert gr.evaluateStaticGraphReconstruction(karate, m, X, None)[0] == pytest.approx(REF['karate_hope_fresh'], abs=1e-12)
    m = HOPE(d=32, beta=0.01)
    X = np.load(golden_path('hope_sbm1024_d32.npz'))['X']
    assert gr.evaluateStaticGraphReconstruction(sbm1024, m, X, None)[0] == pytest.approx(REF['sbm1024_hope_d32'], abs=1e-12)
    m = node2vec(d=2, max_iter=1, walk_len=80, num_walks=10, con_size=10, ret_p=1, inout_p=1)
    X = np.loadtxt(golden_path('ref_karate_node2vec.txt'))
    assert gr.evaluateStaticGraphReconstruction(karate, m, X, None)[0] == pytest.approx(REF['karate_n2v_golden'], abs=1e-12)


def test_reconstructed_adj_equals_pairwise_edge_weight(karate):
    rng = np.random.RandomState(1)
    X = rng.randn(34, 4)
    for m in (HOPE(d=4, beta=0.01), GraphFactorization(d=4, max_iter=1, eta=0.1, regu=0.1)):
        A = m.get_reconstructed_adj(X)
        for i, j in ((0, 1), (5, 3), (33, 0), (7, 7)):
            want = 0.0 if i == j else m.get_edge_weight(i, j)
            assert A[i, j] == pytest.approx(want, rel=1e-12, abs=1e-15)


def test_sampled_map_equals_full(sbm1024):
    g = np.load(golden_path('gf_sbm1024_d32.npz'))
    X = g['X']
    m = GraphFactorization(d=32, max_iter=1, eta=0.02, regu=0.01)
    est = m.get_reconstructed_adj(X)
    truth = gr._adjacency_bool(sbm1024, 1024)
    ap = gr.average_precision_rows(est, truth)
    nodes = [0, 17, 500, 1000, 1023]
    got = gr.sampled_map(sbm1024, lambda i: np.where(np.arange(1024) == i, 0.0, X @ X[i]), nodes)
    assert got == pytest.approx(ap[nodes].mean(), abs=1e-12)


def test_stored_snap_embeddings_carry_the_recorded_map(sbm1024):
    """tests/golden/n2v_snap_sbm1024_*.npz are embeddings written by the real SNAP binary (first run of each setting in
    scripts/make_golden.py); their MAP under this evaluator equals the value the reference evaluator recorded in
    n2v_ref.json, and they show what racing OpenMP threads do to the result (t8: collapsed onto one direction)."""
    ref = json.load(open(golden_path('n2v_ref.json')))
    m = node2vec(d=128, max_iter=1, walk_len=80, num_walks=10, con_size=10, ret_p=1, inout_p=1)
    cos = {}
    for thr in ('t1', 't8'):
        for d in (16, 128):
            X = np.load(golden_path('n2v_snap_sbm1024_d%d_%s.npz' % (d, thr)))['X'].astype(np.float64)
            MAP = gr.evaluateStaticGraphReconstruction(sbm1024, m, X, None)[0]
            assert MAP == pytest.approx(ref['sbm1024_d%d_%s' % (d, thr)][0], abs=3e-4)        # stored as float32
            U = X / np.linalg.norm(X, axis=1, keepdims=True)
            cos[(d, thr)] = float(np.linalg.norm(U.mean(axis=0)))                               # 1.0 = all rows parallel
    # race-free rows share a common component (|mean unit vector| ~0.88-0.89); the 8-thread runs are almost parallel (~0.995)
    assert cos[(16, 't8')] > 0.99 > 0.93 > cos[(16, 't1')] and cos[(128, 't8')] > 0.99 > 0.93 > cos[(128, 't1')]


def test_eligible_sample_and_the_batched_cpu_scorer_behind_the_rmat_goldens():
    """The R-MAT parity goldens (n2v_ref_oracle_rmat*_e16k.json) are per-node APs of the oracle's embedding over reconstruction.eligible_sample -- nodes that
    have a neighbour j > i, the only ones metrics.computeMAP can give a non-zero AP -- computed by scripts/score_oracle_ap.ap_of_nodes (batched, fp64
    scores, exact tie rule).  Pin both: the sample's definition, and the scorer against average_precision_rows (= metrics.computeMAP node by node) on a
    graph with hubs, isolated nodes and tied scores."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'scripts'))
    import score_oracle_ap
    from gem_amd.graph import rmat_graph
    g = rmat_graph(10, 12000, 5)
    nodes = gr.eligible_sample(g, 200)
    has_upper = np.zeros(g.n, bool); has_upper[g.src[g.dst > g.src]] = True
    assert len(nodes) == 200 and np.all(np.diff(nodes) > 0) and has_upper[nodes].all()
    assert np.array_equal(nodes, gr.eligible_sample(g, 200)) and not np.array_equal(nodes, gr.eligible_sample(g, 200, seed=2))
    assert len(gr.eligible_sample(g, 10 ** 6)) == int(has_upper.sum())                      # capped at the eligible nodes
    X = np.random.RandomState(0).randn(g.n, 16).astype(np.float32)
    X[::7] = X[3]                                                                              # ties
    order = np.argsort(g.src, kind='stable')
    got = score_oracle_ap.ap_of_nodes(X, g.dst[order].astype(np.int64), np.searchsorted(g.src[order], np.arange(g.n + 1)), np.arange(g.n))
    X64 = X.astype(np.float64)
    T = np.zeros((g.n, g.n), bool); T[g.src, g.dst] = True
    want = gr.average_precision_rows(X64 @ X64.T, T)
    assert np.abs(got - want).max() < 1e-12 and np.all(got[~has_upper] == 0.0)
