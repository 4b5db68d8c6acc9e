"""GPU parity tests for gf_edge_sgd: HIP path (through the C ABI) vs the CPU oracle
on the same seeded inputs, vs the reference-generated goldens, and size-independent
properties at BASELINE configs[1] (SBM 10k/100k, d=128)."""
import ctypes as C
import json

import numpy as np
import pytest

import oracle
from gem_amd import _hip
from gem_amd.embedding.gf import GraphFactorization
from gem_amd.evaluation import reconstruction as gr
from gem_amd.graph import EdgeListGraph, edge_arrays, sbm_graph
from conftest import golden_path

pytestmark = pytest.mark.gpu

# fp32 device result vs fp32 sequential CPU loop: the only difference is the summation
# order of the d-term dot product (wave tree vs left-to-right) and FMA contraction.
RTOL = 2e-5


def hip_train(n, src, dst, w, d, eta, regu, iters, X0):
    X = np.ascontiguousarray(X0, dtype=np.float32).copy()
    stats = (C.c_double * 4)()
    _hip.check(_hip.lib().gemhip_gf_train(n, len(src), _hip.ptr(_hip.as_i32(src), C.c_int32),
                                          _hip.ptr(_hip.as_i32(dst), C.c_int32), _hip.ptr(_hip.as_f32(w), C.c_float), d,
                                          eta, regu, iters, _hip.ptr(X, C.c_float), stats))
    return X, list(stats)


def assert_close(X, ref):
    scale = max(float(np.abs(ref).max()), 1e-30)
    err = float(np.abs(X.astype(np.float64) - ref.astype(np.float64)).max())
    assert err <= RTOL * scale, 'max|diff|=%.3e vs scale %.3e' % (err, scale)


@pytest.mark.parametrize('d', [2, 7, 32, 64, 128, 130, 256, 512])
def test_karate_matches_oracle_all_widths(karate, d):
    """karate's node insertion order is 0,31,21,19,...: exercises the multi-level schedule."""
    n, src, dst, w, _ = edge_arrays(karate)
    X0 = 0.1 * np.random.RandomState(d).randn(n, d)
    X, stats = hip_train(n, src, dst, w, d, 0.05, 0.01, 25, X0)
    assert stats[3] >= 2                      # more than one dependency level on this ordering
    assert_close(X, oracle.gf_train_f32(n, src, dst, w, d, 0.05, 0.01, 25, X0))


@pytest.mark.parametrize('d,iters', [(32, 5), (128, 3)])
def test_sbm1024_matches_oracle(sbm1024, d, iters):
    n, src, dst, w, _ = edge_arrays(sbm1024)
    X0 = 0.01 * np.random.RandomState(5).randn(n, d)
    X, stats = hip_train(n, src, dst, w, d, 0.02, 0.01, iters, X0)
    assert stats[1] == 21045 and stats[3] == 1
    assert_close(X, oracle.gf_train_f32(n, src, dst, w, d, 0.02, 0.01, iters, X0))


@pytest.mark.parametrize('tag,gname', [('karate_ref_hp', 'karate'), ('karate_train', 'karate'), ('sbm1024_d32', 'sbm1024')])
def test_matches_reference_python_golden(tag, gname, request):
    """Vectors produced by gem/embedding/gf.py itself (fp64); device is fp32."""
    g = np.load(golden_path('gf_%s.npz' % tag))
    n, src, dst, w, _ = edge_arrays(request.getfixturevalue(gname))
    X, _ = hip_train(n, src, dst, w, int(g['d']), float(g['eta']), float(g['regu']), int(g['max_iter']), g['X0'])
    scale = float(np.abs(g['X']).max())
    assert float(np.abs(X - g['X']).max()) <= 1e-4 * scale


def test_weighted_shuffled_order_graph():
    """Random weights, random node insertion order, hub row longer than one wave chunk."""
    rng = np.random.RandomState(9)
    n = 300
    src = rng.randint(0, n, 4000); dst = rng.randint(0, n, 4000)
    hub = np.full(200, 3); hub_dst = rng.permutation(np.arange(4, n))[:200]
    src = np.concatenate([src, hub]); dst = np.concatenate([dst, hub_dst])
    keep = src != dst
    key = np.unique(src[keep].astype(np.int64) * n + dst[keep])
    src, dst = (key // n).astype(np.int32), (key % n).astype(np.int32)
    # reference iteration order = grouped by source in a shuffled node order
    order = rng.permutation(n); rank = np.empty(n, int); rank[order] = np.arange(n)
    perm = np.lexsort((rng.rand(len(src)), rank[src]))
    src, dst = src[perm], dst[perm]
    w = rng.rand(len(src)).astype(np.float32) * 2
    X0 = 0.1 * rng.randn(n, 16)
    X, stats = hip_train(n, src, dst, w, 16, 0.05, 0.02, 10, X0)
    assert stats[3] > 3
    assert_close(X, oracle.gf_train_f32(n, src, dst, w, 16, 0.05, 0.02, 10, X0))


def test_edge_cases():
    X0 = np.random.RandomState(1).randn(5, 4).astype(np.float32)
    e = np.zeros(0, np.int32)
    X, _ = hip_train(5, e, e, None, 4, 0.1, 0.1, 3, X0)              # empty edge list
    assert np.array_equal(X, X0)
    X, _ = hip_train(5, np.array([4, 3, 2, 2]), np.array([0, 1, 2, 1]), None, 4, 0.1, 0.1, 3, X0)   # nothing fires (j<=i)
    assert np.array_equal(X, X0)
    X, _ = hip_train(5, np.array([0, 1]), np.array([1, 2]), None, 4, 0.1, 0.1, 0, X0)   # max_iter = 0
    assert np.array_equal(X, X0)
    with pytest.raises(_hip.GemHipError):
        hip_train(5, np.array([0]), np.array([9]), None, 4, 0.1, 0.1, 1, X0)             # endpoint out of range


def test_objective_kernel(sbm1024):
    n, src, dst, w, _ = edge_arrays(sbm1024)
    X = (0.3 * np.random.RandomState(2).randn(n, 128)).astype(np.float32)
    out = (C.c_double * 2)()
    _hip.check(_hip.lib().gemhip_gf_objective(n, len(src), _hip.ptr(src, C.c_int32), _hip.ptr(dst, C.c_int32),
                                              _hip.ptr(w, C.c_float), 128, _hip.ptr(X, C.c_float), out))
    f1, f2 = oracle.gf_objective(n, src, dst, w, 128, X)
    assert out[0] == pytest.approx(f1, rel=1e-5) and out[1] == pytest.approx(f2, rel=1e-5)


def test_class_api_and_map_parity(karate, sbm1024):
    """GraphFactorization.learn_embedding through the plugin API: same init stream as the
    reference (np.random.seed), MAP equal to the reference run's MAP (map_ref.json)."""
    ref = json.load(open(golden_path('map_ref.json')))
    for G, tag, key in ((karate, 'karate_train', 'karate_gf_train'), (sbm1024, 'sbm1024_d32', 'sbm1024_gf_d32_5sweeps')):
        g = np.load(golden_path('gf_%s.npz' % tag))
        np.random.seed(int(g['seed']))
        m = GraphFactorization(d=int(g['d']), max_iter=int(g['max_iter']), eta=float(g['eta']), regu=float(g['regu']),
                               data_set='t')
        Y = m.learn_embedding(graph=G, edge_f=None, is_weighted=True, no_python=True)
        assert Y.dtype == np.float64 and Y.shape == g['X'].shape and m.get_embedding() is Y
        assert float(np.abs(Y - g['X']).max()) <= 1e-4 * float(np.abs(g['X']).max())
        MAP = gr.evaluateStaticGraphReconstruction(G, m, Y, None)[0]
        assert abs(MAP - ref[key]) <= 0.01 * ref[key]


def test_baseline_config_properties():
    """BASELINE configs[1]: SBM 10k nodes / 100k edges, d=128 -- size-independent properties."""
    g = sbm_graph(10000, 100000, 10, seed=20260924)
    n, src, dst, w, _ = edge_arrays(g)
    X0 = (0.01 * np.random.RandomState(0).randn(n, 128)).astype(np.float32)
    Xa, stats = hip_train(n, src, dst, w, 128, 1e-2, 1e-2, 40, X0)
    Xb, _ = hip_train(n, src, dst, w, 128, 1e-2, 1e-2, 40, X0)
    assert np.array_equal(Xa, Xb)                                     # deterministic, schedule-independent
    fires = np.zeros(n, bool); fires[src[dst > src]] = True
    assert np.array_equal(Xa[~fires], X0[~fires])                     # rows with no j>i edge keep their init (gf.py:95-96)
    assert stats[1] == int((dst > src).sum())
    # composition: 25 + 15 sweeps through the staged API == 40 sweeps
    L = _hip.lib(); plan = C.c_void_p()
    _hip.check(L.gemhip_gf_plan_create(n, len(src), _hip.ptr(src, C.c_int32), _hip.ptr(dst, C.c_int32), None, 128, 0, n,
                                       C.byref(plan)))
    _hip.check(L.gemhip_gf_plan_set_embedding(plan, _hip.ptr(X0, C.c_float)))
    _hip.check(L.gemhip_gf_plan_sweeps(plan, 25, 1e-2, 1e-2, None))
    _hip.check(L.gemhip_gf_plan_sweeps(plan, 15, 1e-2, 1e-2, None))
    Xc = np.empty_like(X0)
    _hip.check(L.gemhip_gf_plan_get_embedding(plan, _hip.ptr(Xc, C.c_float)))
    _hip.check(L.gemhip_gf_plan_destroy(plan))
    assert np.array_equal(Xc, Xa)
    # the objective the reference prints (gf.cpp:94-113) goes down, and equals the oracle's trajectory end point
    f0 = sum(oracle.gf_objective(n, src, dst, w, 128, X0)); f1 = sum(oracle.gf_objective(n, src, dst, w, 128, Xa))
    assert f1 < f0
    assert_close(Xa, oracle.gf_train_f32(n, src, dst, w, 128, 1e-2, 1e-2, 40, X0))


def test_device_init_distribution():
    L = _hip.lib(); plan = C.c_void_p()
    e = np.array([0], np.int32); f = np.array([1], np.int32)
    _hip.check(L.gemhip_gf_plan_create(4096, 1, _hip.ptr(e, C.c_int32), _hip.ptr(f, C.c_int32), None, 128, 0, 4096, C.byref(plan)))
    _hip.check(L.gemhip_gf_plan_init_embedding(plan, 42, 0.01))
    X = np.empty((4096, 128), np.float32)
    _hip.check(L.gemhip_gf_plan_get_embedding(plan, _hip.ptr(X, C.c_float)))
    _hip.check(L.gemhip_gf_plan_destroy(plan))
    assert abs(X.mean()) < 1e-4 and abs(X.std() - 0.01) < 1e-4
    assert abs(float(((X / 0.01) ** 4).mean()) - 3.0) < 0.05             # gaussian kurtosis


def test_device_init_option_trains_like_the_numpy_init(sbm1024):
    """device_init=True: same algorithm from a Philox-drawn 0.01*N(0,1) table (what gf.cpp does with its own generator)."""
    from gem_amd.evaluation import reconstruction as gr
    maps = {}
    for dev in (False, True):
        m = GraphFactorization(d=32, max_iter=60, eta=0.02, regu=0.01, seed=3, device_init=dev)
        X = m.learn_embedding(graph=sbm1024, is_weighted=True, no_python=True)
        assert X.shape == (1024, 32) and X.dtype == np.float64 and np.isfinite(X).all()
        assert m._stats['levels'] == 1 and m._stats['updates_per_sweep'] == 21045
        maps[dev] = gr.evaluateStaticGraphReconstruction(sbm1024, m, X, None)[0]
    assert abs(maps[True] - maps[False]) < 0.25 * maps[False], maps          # two different random inits
    a = GraphFactorization(d=32, max_iter=2, eta=0.02, regu=0.01, seed=3, device_init=True).learn_embedding(graph=sbm1024)
    b = GraphFactorization(d=32, max_iter=2, eta=0.02, regu=0.01, seed=3, device_init=True).learn_embedding(graph=sbm1024)
    assert np.array_equal(a, b)                                   # seeded and deterministic


@pytest.mark.parametrize('d', [2, 7, 128, 256, 1024])
def test_hub_kernel_is_bit_identical_to_the_wave_per_row_kernel(d, sbm1024, karate, monkeypatch):
    """Rows with many firing edges get a workgroup (producers stream neighbour rows through LDS, one wavefront applies the chain):
    same gf_apply_edge in the same order, so the tables are bit-identical to the wave-per-row kernel's.  The threshold is lowered
    so that most rows of the test graphs take the hub kernel (partial batches, multi-level schedules, every row layout)."""
    for G, sweeps in ((sbm1024, 3), (karate, 5)):
        n, src, dst, w, _ = edge_arrays(G)
        np.random.seed(3)
        X0 = (0.01 * np.random.randn(n, d)).astype(np.float32)
        out = {}
        for tag, env in (('hub', {'GEMHIP_GF_HUB_EDGES': '3'}), ('plain', {'GEMHIP_GF_NO_HUB_KERNEL': '1'})):
            monkeypatch.delenv('GEMHIP_GF_HUB_EDGES', raising=False); monkeypatch.delenv('GEMHIP_GF_NO_HUB_KERNEL', raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            out[tag] = hip_train(n, src, dst, w, d, 0.01, 0.01, sweeps, X0)[0]
        assert np.array_equal(out['hub'], out['plain'])


@pytest.mark.parametrize('d', [7, 128, 256])
@pytest.mark.parametrize('k', [2, 5, 8, 64])
def test_rows_per_wave_kernel_is_bit_identical(d, k, sbm1024, karate):
    """gf_sweep_rows_kernel (K consecutive rows per wavefront, the next row's (col, w) chunk and X_i in flight while a row is trained: what large
    levels run) applies the same gf_apply_edge in the same edge order as gf_sweep_kernel: bit-identical tables for every K, including K that does not
    divide the level, multi-level schedules (karate) and rows of more than one 64-edge chunk (the dense graph)."""
    rs = np.random.RandomState(9)
    dense_src = np.repeat(np.arange(40, dtype=np.int32), 200); dense_dst = rs.randint(0, 400, dense_src.size).astype(np.int32)     # 200 edges per source row
    for name, (n, src, dst, w) in (('sbm1024', edge_arrays(sbm1024)[:4]), ('karate', edge_arrays(karate)[:4]), ('dense', (400, dense_src, dense_dst, None))):
        X0 = (0.01 * np.random.RandomState(3).randn(n, d)).astype(np.float32)
        L = _hip.lib()
        out = {}
        for kk in (1, k):
            plan = C.c_void_p()
            _hip.check(L.gemhip_gf_plan_create(n, len(src), _hip.ptr(_hip.as_i32(src), C.c_int32), _hip.ptr(_hip.as_i32(dst), C.c_int32),
                                               _hip.ptr(_hip.as_f32(w), C.c_float), d, 0, n, C.byref(plan)))
            _hip.check(L.gemhip_gf_plan_set_embedding(plan, _hip.ptr(X0, C.c_float)))
            _hip.check(L.gemhip_gf_plan_set_rows_per_wave(plan, kk))
            _hip.check(L.gemhip_gf_plan_sweeps(plan, 4, 0.02, 0.01, None))
            X = np.empty_like(X0)
            _hip.check(L.gemhip_gf_plan_get_embedding(plan, _hip.ptr(X, C.c_float)))
            _hip.check(L.gemhip_gf_plan_destroy(plan))
            out[kk] = X
        assert np.array_equal(out[1], out[k]), name
        if name != 'dense':
            assert_close(out[k], oracle.gf_train_f32(n, src, dst, w, d, 0.02, 0.01, 4, X0))


@pytest.mark.parametrize('d', [7, 128, 256])
@pytest.mark.parametrize('k,grid', [(2, 0), (5, 8), (64, 0)])
def test_fused_sweeps_in_one_cooperative_launch_are_bit_identical(d, k, grid, sbm1024, karate):
    """gemhip_gf_plan_set_fused_sweeps: up to k sweeps per COOPERATIVE launch (gf_sweeps_coop_kernel: a resident grid, the sweeps separated by a grid barrier
    with agent-scope release / acquire instead of a kernel boundary) -- same row body, same edge order: the tables equal the launch loop's bit for bit, for
    sweep counts that k does not divide (13), odd k (the current table flips), a grid smaller than the level (grid-stride rows), and on a multi-level plan
    (karate), which keeps the launch loop."""
    for name, G in (('sbm1024', sbm1024), ('karate', karate)):
        n, src, dst, w, _ = edge_arrays(G)
        X0 = (0.01 * np.random.RandomState(3).randn(n, d)).astype(np.float32)
        L = _hip.lib()
        out = {}
        for kk in (0, k):
            plan = C.c_void_p()
            _hip.check(L.gemhip_gf_plan_create(n, len(src), _hip.ptr(_hip.as_i32(src), C.c_int32), _hip.ptr(_hip.as_i32(dst), C.c_int32),
                                               _hip.ptr(_hip.as_f32(w), C.c_float), d, 0, n, C.byref(plan)))
            _hip.check(L.gemhip_gf_plan_set_embedding(plan, _hip.ptr(X0, C.c_float)))
            _hip.check(L.gemhip_gf_plan_set_fused_sweeps(plan, kk, grid))
            _hip.check(L.gemhip_gf_plan_sweeps(plan, 13, 0.02, 0.01, None))
            _hip.check(L.gemhip_gf_plan_sweeps(plan, 1, 0.02, 0.01, None))
            X = np.empty_like(X0)
            _hip.check(L.gemhip_gf_plan_get_embedding(plan, _hip.ptr(X, C.c_float)))
            _hip.check(L.gemhip_gf_plan_destroy(plan))
            out[kk] = X
        assert np.array_equal(out[0], out[k]), name
        assert_close(out[k], oracle.gf_train_f32(n, src, dst, w, d, 0.02, 0.01, 14, X0))


@pytest.mark.parametrize('case', ['karate', 'sbm1024'])
def test_hip_equals_the_emb_file_the_gf_cpp_binary_wrote(case, request):
    """The reference's NATIVE path end to end: gf.cpp's binary, run deterministically (frozen clock, scripts/make_golden_gf_cpp.py), wrote
    tests/golden/gf_cpp_binary_<case>.emb; from the same initial table (gf.cpp:41-52 restated, pinned to the binary on the CPU tier) the HIP sweeps
    land on that file within the fp32 dot-product tolerance plus half a unit of its sixth printed digit."""
    meta = json.load(open(golden_path('gf_cpp_binary.json')))[case]
    G = request.getfixturevalue(meta['graph'])
    n, src, dst, w, _ = edge_arrays(G)
    X0 = oracle.gf_cpp_init(meta['seed32'], n, meta['d'])
    X, _ = hip_train(n, src, dst, None, meta['d'], meta['eta'], meta['regu'], meta['max_iter'], X0)
    want = np.loadtxt(golden_path('gf_cpp_binary_%s.emb' % case), skiprows=1)[:, 1:]
    scale = float(np.abs(want).max())
    assert float(np.abs(X - want).max()) <= RTOL * scale + 0.5e-5 * scale
