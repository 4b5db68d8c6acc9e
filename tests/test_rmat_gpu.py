"""GPU parity on a power-law (R-MAT) graph -- hub rows far longer than a wavefront chunk, the shape of
BASELINE configs[4] (R-MAT scale-22) at test size."""
import ctypes as C

import numpy as np
import pytest

import oracle
from gem_amd import _hip
from gem_amd.graph import edge_arrays, rmat_graph
from test_gf_gpu import hip_train, assert_close
from test_n2v_gpu import Dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def rmat():
    g = rmat_graph(13, 160000, seed=20260928)
    n, src, dst, w, _ = edge_arrays(g)
    deg = np.bincount(src, minlength=n)
    assert deg.max() > 500 and (deg == 0).sum() > 100          # hubs and isolated nodes
    return g


def test_gf_on_hub_rows(rmat):
    n, src, dst, w, _ = edge_arrays(rmat)
    X0 = 0.05 * np.random.RandomState(1).randn(n, 128)
    X, stats = hip_train(n, src, dst, w, 128, 0.002, 0.01, 4, X0)
    assert_close(X, oracle.gf_train_f32(n, src, dst, w, 128, 0.002, 0.01, 4, X0))


def test_walks_weighted_second_order_on_hubs(rmat):
    n, src, dst, _, _ = edge_arrays(rmat)
    w = (np.random.RandomState(2).rand(len(src)) + 0.25).astype(np.float32)
    dev = Dev(n, src, dst, w)
    row_ptr, col, ww = oracle.sorted_csr(n, src, dst, w)
    U, K = oracle.n2v_alias_rows(row_ptr, ww)
    for p, q in ((1.0, 1.0), (0.5, 2.0)):
        got = dev.walks(p, q, 2, 40, 17, 11)
        assert np.array_equal(got, oracle.n2v_walks(row_ptr, col, U, K, p, q, 2, 40, 17, 11))
    c, UT, KT = dev.unigram()
    assert np.array_equal(c, oracle.n2v_vocab(n, got))
    dev.close()


@pytest.mark.hogwild_stat
def test_node2vec_hogwild_map_on_power_law_graph(rmat):
    """Hub rows are the contended ones under Hogwild: the GPU path must still land on the sequential oracle's MAP."""
    from gem_amd.embedding.node2vec import node2vec
    from gem_amd.evaluation import reconstruction as gr
    n, src, dst, w, _ = edge_arrays(rmat)
    maps = []
    for seed in (1, 2):
        m = node2vec(d=16, max_iter=1, walk_len=80, num_walks=10, con_size=10, ret_p=1, inout_p=1, seed=seed)
        maps.append(gr.evaluateStaticGraphReconstruction(rmat, m, m.learn_embedding(graph=rmat), None)[0])
    # the sequential oracle on the SAME flags as the plugin default (27: the binary's quirks and its unigram-table layout).  On a power-law graph the
    # layout is not cosmetic: under RndUnigramInt's quirk the negative distribution is a function of the alias structure, and the binary's
    # first-appearance layout gives MAP 0.0119 / 0.0125 here (seeds 1 / 2) where the node-id layout of rounds 1-3 gives 0.0173 / 0.0188
    X, _ = oracle.n2v_train(n, src, dst, None, 16, 80, 10, 10, 1, 1.0, 1.0, 1, _hip.N2V_SNAP_LAYOUT)
    ref = gr.evaluateStaticGraphReconstruction(rmat, m, X.astype(np.float64), None)[0]
    assert abs(np.mean(maps) - ref) <= 0.15 * ref, (maps, ref)          # MAP ~0.012: a handful of rank swaps is several percent
    # ... and the node-id layout against ITS oracle
    m11 = node2vec(d=16, max_iter=1, walk_len=80, num_walks=10, con_size=10, ret_p=1, inout_p=1, seed=1, flags=_hip.N2V_SNAP_COMPAT)
    got = gr.evaluateStaticGraphReconstruction(rmat, m11, m11.learn_embedding(graph=rmat), None)[0]
    X11, _ = oracle.n2v_train(n, src, dst, None, 16, 80, 10, 10, 1, 1.0, 1.0, 1, _hip.N2V_SNAP_COMPAT)
    ref11 = gr.evaluateStaticGraphReconstruction(rmat, m11, X11.astype(np.float64), None)[0]
    assert abs(got - ref11) <= 0.15 * ref11 and ref11 > 1.2 * ref, (got, ref11, ref)


@pytest.mark.hogwild_stat
@pytest.mark.parametrize('layout', ['node_id', 'vocab_order'])
@pytest.mark.parametrize('scale', [17, 20, 22])
def test_rmat_default_concurrency_lands_on_the_sequential_oracle(scale, layout):
    """The Hogwild defaults on a SECOND graph family: R-MAT scale 17 (131 072 nodes, 1.86 M edges, max degree 9 510, the top hub 0.5 % of all tokens), scale
    20 (1 048 576 nodes, 15.4 M edges, 432 M tokens) and scale 22 (BASELINE configs[4]: 4.2 M nodes, 62 M edges, 1.59 G tokens; golden made in round 6,
    skipped until it is committed) against the sequential oracle's run on the same seed AND the same unigram-table layout: `node_id` = flags 11,
    `vocab_order` = flags 27, the plugin default (the binary's layout).  ONE launch each.

    The statistic.  metrics.computeMAP ranks the candidates j > i only, so a uniformly sampled node of a power-law graph scores 0 by construction two times
    out of three; the goldens hold the oracle's AP for nodes drawn from those that HAVE a ranked neighbour (reconstruction.eligible_sample: 16 384 of them
    on scale 17 -- their APs sum to 265 --, 131 072 on scale 20 / 22), paired per node: s.e. 0.3-0.5 % per launch.
    What the gap is made of (round 6; profiles/r06_*.jsonl).  (1) A heavy tail, now fixed: the two rows of an isolated edge make up every
    token of their 20 walks and sat in the LDS windows of two wavefronts at once for a whole walk whenever two of those walks were in flight together
    (probability ~ width); both walks' worth of updates were added and the pair ended with 14 % more norm than the ~80 other pairs of the graph, all nearly
    collinear (cosine 0.93), whose true neighbour it then outranked: 2-5 % of the graph's MAP per event, launches at one width anywhere between -2 and -8 %.
    Such nodes are now LOCALLY HOT (tokens per walk that contains them >= 8: never cached); launches at one width agree to 0.2 %.  (2) A smooth rest, not
    understood: -0.45 / -1.6 / -1.8 % at 256 / 768 / 1 536 wavefronts on scale 17, -1.3 / -3.0 / -6.0 % on scale 20, the same in both layouts; it is NOT
    the staleness of the hubs' gradients (fresh reads cut it 4x and change nothing), NOT lost updates (all-atomic writes are no better), NOT too few hot
    rows (a lower threshold is worse).  The planner keeps round 5's concurrent-touch bound with a floor of 256 wavefronts: 256 on scale 17 and 20, 548 on
    scale 22.  Bars: 1.5 % on scale 17 (expected -0.45 %), 2.5 % on scale 20 (-1.3 %: north_star's 1 % is not met there) and on scale 22."""
    import json, os
    from conftest import golden_path
    from gem_amd.evaluation import reconstruction as gr
    # scale 17: 16 384 eligible nodes (their APs sum to 288: s.e. 0.3 %); scale 20: 131 072 (sum 384; over 16 384 the APs sum to 49 and one launch has an s.e. of 1.4 %)
    path = golden_path('n2v_ref_oracle_rmat%d%s_%s.json' % (scale, '' if layout == 'node_id' else '_vocab_order', 'e16k' if scale == 17 else 'e128k'))
    if scale == 22 and not os.path.exists(path):          # (round 6 scored the scale-22 oracle runs over the 16 384-node sub-sample: the 131 072-node scoring did not fit the session)
        path = golden_path('n2v_ref_oracle_rmat22%s_e16k.json' % ('' if layout == 'node_id' else '_vocab_order'))
    if scale == 22 and os.environ.get('GEM_TEST_RMAT22', '1') == '0':
        pytest.skip('GEM_TEST_RMAT22=0')
    if not os.path.exists(path):
        pytest.skip('%s not generated (scripts/make_golden_n2v_scale.py --rmat-scale %d --engine oracle --eligible-sample 16384)' % (os.path.basename(path), scale))
    ref = json.load(open(path))
    pr = ref['params']
    flags = _hip.N2V_SNAP_COMPAT if layout == 'node_id' else _hip.N2V_SNAP_LAYOUT
    assert pr['flags'] == flags
    g = rmat_graph(pr['rmat_scale'], pr['edges'], pr['seed'])
    nodes = gr.eligible_sample(g, len(ref['ap']))
    from gem_amd.embedding.node2vec import node2vec
    m = node2vec(d=pr['d'], max_iter=1, walk_len=pr['walk_len'], num_walks=pr['num_walks'], con_size=pr['window'], ret_p=1, inout_p=1, seed=20260923, flags=flags)
    ap = gr.sampled_ap_gpu(g, None, m.learn_embedding(graph=g, is_weighted=True, no_python=True), nodes)
    d = ap - np.asarray(ref['ap'])
    gap, se = float(d.mean() / ref['MAP']), float(d.std(ddof=1) / np.sqrt(len(d)) / ref['MAP'])
    from conftest import record_stat
    bar = 0.015 if scale == 17 else 0.025 if len(ref['ap']) > 20000 else 0.04          # (a 16 384-node sample of scale 22 has an s.e. of ~1 % per launch)
    # (measured at the planner's 256 wavefronts, round 6: -0.45 % (s.e. 0.3 %) on scale 17, -1.1 / -1.5 % (s.e. 0.4 %) on scale 20)
    record_stat('R-MAT scale %d, %s layout, one Hogwild launch against the sequential oracle (paired, %d nodes)' % (scale, layout, len(d)),
                '%+.2f %% (s.e. %.2f %%)' % (100 * gap, 100 * se), '+-%.1f %%' % (100 * bar))
    assert abs(gap) <= bar, (gap, se, ap.mean(), ref['MAP'])
