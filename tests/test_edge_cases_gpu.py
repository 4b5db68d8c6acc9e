"""GPU edge cases through the plugin API: tiny graphs, isolated nodes, self loops, sinks, minimal dimensions."""
import numpy as np
import pytest

import oracle
from oracle import hope_oracle
from gem_amd import _hip
from gem_amd.embedding.gf import GraphFactorization
from gem_amd.embedding.hope import HOPE
from gem_amd.embedding.node2vec import node2vec
from gem_amd.graph import EdgeListGraph, edge_arrays

pytestmark = pytest.mark.gpu


def tiny():
    # 6 nodes: 0-1-2 triangle-ish, self loop on 2, node 4 isolated, node 5 a sink, duplicate-free, unsorted edge order
    src = np.array([2, 0, 1, 0, 2, 3, 1, 3], np.int32)
    dst = np.array([0, 1, 2, 2, 2, 5, 0, 1], np.int32)
    w = np.array([1.0, 2.0, 0.5, 1.5, 3.0, 1.0, 2.0, 0.25], np.float32)
    return EdgeListGraph(6, src, dst, w)


def test_gf_tiny_and_d1():
    g = tiny()
    n, src, dst, w, _ = edge_arrays(g)
    for d in (1, 3):
        np.random.seed(5)
        X0 = (0.01 * np.random.randn(n, d)).astype(np.float32)
        np.random.seed(5)
        Y = GraphFactorization(d=d, eta=0.05, regu=0.1, max_iter=30).learn_embedding(graph=g)
        ref = oracle.gf_train_f32(n, src, dst, w, d, 0.05, 0.1, 30, X0)
        assert np.abs(Y - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1e-3)
        assert np.array_equal(Y[4], X0[4].astype(np.float64))                 # isolated node keeps its init


def test_hope_tiny_k1():
    g = tiny()
    n, src, dst, w, _ = edge_arrays(g)
    m = HOPE(d=2, beta=0.05)
    Y = m.learn_embedding(graph=g)
    Xo, so = hope_oracle.hope_dense(hope_oracle.adjacency(n, src, dst, w), 0.05, 2)
    assert np.allclose(m._sigma, so, rtol=1e-4)
    assert np.allclose(np.abs(Y), np.abs(Xo), atol=2e-5)
    with pytest.raises(ValueError):
        HOPE(d=12, beta=0.05).learn_embedding(graph=g)                        # k = 6 >= n: svds would refuse too
    with pytest.raises(ValueError):
        HOPE(d=1, beta=0.05).learn_embedding(graph=g)                         # k = 0


def test_node2vec_tiny_with_sink_and_isolated_nodes():
    g = tiny()
    n = 6
    for flags in (11, 8):                                                     # SNAP-compatible padding, and clean padding
        m = node2vec(d=4, max_iter=2, walk_len=7, num_walks=3, con_size=2, ret_p=0.5, inout_p=2.0, seed=1, flags=flags)
        Y = m.learn_embedding(graph=g, is_weighted=True)
        assert Y.shape == (n, 4) and np.isfinite(Y).all()
        node2vec.hyper_params.pop('flags', None)
    # walk_len 1: no contexts at all -> embedding == InitPosEmb
    m = node2vec(d=4, max_iter=1, walk_len=1, num_walks=2, con_size=3, ret_p=1, inout_p=1, seed=9)
    Y = m.learn_embedding(graph=g)
    P0, _ = oracle.sgns_init(n, 4, 9)
    assert np.array_equal(Y.astype(np.float32), P0)


def test_bad_arguments_are_reported_not_crashed():
    import ctypes as C
    L = _hip.lib()
    X = np.zeros((3, 2), np.float32)
    rc = L.gemhip_gf_train(3, 1, _hip.ptr(np.array([0], np.int32), C.c_int32), _hip.ptr(np.array([7], np.int32), C.c_int32), None, 2, 0.1, 0.1, 1,
                           _hip.ptr(X, C.c_float), None)
    assert rc == -1 and b'outside' in L.gemhip_last_error()
    rp = np.array([0, 1, 1, 2], np.int64); col = np.array([1, 5], np.int32)
    h = C.c_void_p()
    assert L.gemhip_n2v_create(3, 2, _hip.ptr(rp, C.c_int64), _hip.ptr(col, C.c_int32), None, C.byref(h)) == -1
    assert L.gemhip_hope(3, 2, _hip.ptr(rp, C.c_int64), _hip.ptr(np.array([1, 0], np.int32), C.c_int32), None, 0.01, 5, 2, 3, 2, 1e-5, 1,
                         _hip.ptr(X, C.c_float), _hip.ptr(X, C.c_float), _hip.ptr(X, C.c_float), None) == -1


@pytest.mark.parametrize('d', [384, 512])
def test_node2vec_wide_rows_fall_back_to_the_round1_kernel(d):
    """The LDS window of the default SGNS kernel holds 2R+1 rows (twice with the delta write-back): at d >= 384 it no longer fits a block's
    64 KB and training runs on the round-1 kernel instead of failing."""
    from gem_amd.graph import sbm_graph
    g = sbm_graph(2048, 20480, 2, seed=3)
    m = node2vec(d=d, max_iter=1, walk_len=20, num_walks=2, con_size=10, ret_p=1, inout_p=1, seed=1)
    Y = m.learn_embedding(graph=g, is_weighted=True, no_python=True)
    assert Y.shape == (2048, d) and np.isfinite(Y).all() and np.abs(Y).max() > 1e-4
