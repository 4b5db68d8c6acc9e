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
def test_rmat17_default_concurrency_lands_on_the_sequential_oracle(layout):
    """The Hogwild defaults on a SECOND graph family at >= scale 17: R-MAT scale 17 -- 131 072 nodes, 1.86 M edges, max degree 9 510,
    effective table size of the negative-sampling distribution 11 316 -- against the sequential oracle's run on the same seed AND the same unigram-table
    layout (per-node APs paired over 2 048 sampled nodes): `node_id` = flags 11 against tests/golden/n2v_ref_oracle_rmat17.json (1 466 s of CPU; rounds 2-3),
    `vocab_order` = flags 27, the plugin default since round 4 (the binary's layout), against n2v_ref_oracle_rmat17_vocab_order.json.  Round 2's setting
    (1024 wavefronts, every context row cached) lost 15-17 % of the MAP here; with hot rows kept out of the LDS windows and the wavefront count from the
    effective table size the gap measured +0.5 +- 0.6 % and +0.7 +- 0.5 % (profiles/r03_rmat17_rule_check.jsonl).  Bar 3 % on the mean of up to three launches."""
    import json, os
    from conftest import golden_path
    from gem_amd.evaluation import reconstruction as gr
    path = golden_path('n2v_ref_oracle_rmat17.json' if layout == 'node_id' else 'n2v_ref_oracle_rmat17_vocab_order.json')
    if not os.path.exists(path):
        pytest.skip('golden %s not generated yet (scripts/make_golden_n2v_scale.py --flags 27 --rmat-scale 17)' % os.path.basename(path))
    ref = json.load(open(path))
    pr = ref['params']
    flags = _hip.N2V_SNAP_COMPAT if layout == 'node_id' else _hip.N2V_SNAP_LAYOUT
    assert pr.get('flags', 11) == flags
    g = rmat_graph(pr['rmat_scale'], pr['edges'], pr['seed'])
    nodes = np.random.RandomState(0).choice(g.n, size=len(ref['ap']), replace=False)
    from gem_amd.embedding.node2vec import node2vec
    m = node2vec(d=pr['d'], max_iter=1, walk_len=pr['walk_len'], num_walks=pr['num_walks'], con_size=pr['window'], ret_p=1, inout_p=1, seed=20260923, flags=flags)
    # A Hogwild launch is not deterministic: on this graph (MAP 0.0055 -- a handful of rank swaps is a percent) the same seed measured -2.8 ... +1.4 % over
    # six launches (DESIGN_NOTES.md; mean -0.4 %, run-to-run s.d. ~1.5 %), so ONE launch against a 3 % bar is a 2-sigma test and did fail once in round 4
    # (-3.4 %).  One launch inside 2 % passes; otherwise the statement tested is the one the notes make -- the MEAN of three launches inside 3 %.
    gaps = []
    for attempt in range(3):
        ap = gr.sampled_ap_gpu(g, None, m.learn_embedding(graph=g, is_weighted=True, no_python=True), nodes)
        gaps.append(float((ap - np.asarray(ref['ap'])).mean() / ref['MAP']))
        if attempt == 0 and abs(gaps[0]) <= 0.02:
            break
    assert abs(np.mean(gaps)) <= 0.03, (gaps, ap.mean(), ref['MAP'])
