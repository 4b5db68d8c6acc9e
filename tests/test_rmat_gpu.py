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
@pytest.mark.parametrize('scale', [17, 20])
def test_rmat_default_concurrency_lands_on_the_sequential_oracle(scale, layout):
    """The Hogwild defaults on a SECOND graph family at >= scale 17: R-MAT scale 17 -- 131 072 nodes, 1.86 M edges, max degree 9 510, the top hub 0.5 % of
    all tokens -- against the sequential oracle's run on the same seed AND the same unigram-table layout: `node_id` = flags 11, `vocab_order` = flags 27,
    the plugin default (the binary's layout).  ONE launch, bar 3 %.

    The statistic (round 5).  Rounds 2-4 paired per-node APs over 2 048 uniformly sampled nodes (n2v_ref_oracle_rmat17*.json).  metrics.computeMAP
    ranks the candidates j > i only, so 1 362 of those nodes score 0 by construction and the other 686 sum to 11: one node whose only neighbour lands
    on rank 1 moves that "gap" by 9 % (round 4's driver run: +2.1, +7.7, +8.8 % on one box; -3.4 % on another).  The goldens read here
    (n2v_ref_oracle_rmat17*_e16k.json, scripts/make_golden_n2v_scale.py --eligible-sample 16384 on the same 1 480 s oracle runs) hold the oracle's AP
    for 16 384 nodes drawn from the 44 073 that HAVE a ranked neighbour (reconstruction.eligible_sample): the paired gap of one launch has a
    standard error of 0.3 %.
    The rule.  With that statistic round 3's launch rule (602 wavefronts here) measured -3.7 % (vocab order) / -3.4 % (node id) over nine launches on
    two boxes, launch to launch anywhere between +0.9 and -6.7 %: the hubs' atomic updates carry gradients computed one to two pair steps earlier, and
    with W wavefronts (W - 1) x s x sum (p_v + 5 q_v)^2 / 6 others touch the same row inside that window.  The planner now bounds that number at 0.2
    (n2v.hip plan_sgns_launch; at 155 wavefronts: -0.41 % / +0.14 % over 24 launches on two boxes, profiles/r05_rmat17_launches.jsonl; the sweep measured
    -0.1 ... -0.9 % at 128 - 256 wavefronts, profiles/r05_rmat17_width_sweep.jsonl).
    Scale 20 (1 048 576 nodes, 15.4 M edges, 432 M tokens; 3.7 h of CPU per oracle run, scripts/oracle_n2v_resumable.py): the same family three doublings up
    -- the largest power-law graph the sequential oracle has been run on (BASELINE configs[4] is scale 22) -- and MORE sensitive at the same width: with
    the bound as first calibrated on scale 17 alone (688 wavefronts) it measured -6.4 % (binary's layout) / -8.3 % (node id), at 256 wavefronts -1.9 %
    (profiles/r05_rmat20_launches_e128k.jsonl; paired over 131 072 eligible nodes, s.e. 0.4 %).  The bound was tightened to the worse graph ((W - 1) x
    touch2_hub <= 0.165: 207 wavefronts here, 50 on scale 17): the default layout then measures -1.51 % (s.e. 0.34 %).  north_star's 1 % is NOT met at
    scale 20; the bar here is 5 %.
    The node-id layout at scale 20 measured -6.29 % (s.e. 0.92 %) at those 207 wavefronts (profiles/r05_pytest_gpu_final2_scale20_node_id_failed.log) and
    -1.50 % (s.e. 0.64 %) at 104 (profiles/r05_rmat20_node_id_104_wavefronts.log): the two layouts are different SAMPLERS under RndUnigramInt's quirk
    (only alias targets are ever drawn, so the distribution follows Vose's pairing, i.e. the table order: the oracle's own MAP differs by 18.9 % between
    them here, 8.7 % on scale 17, +8.0 ... +9.4 % over four seeds on scale 14 with a seed-to-seed s.d. of 0.6 %: profiles/r05_oracle_layout_vs_seed_rmat14.json),
    and none of the statistics of the sampled distribution that were tried orders their sensitivity on both graphs
    (profiles/r05_negative_distribution_by_layout.json).  The planner therefore halves the bound for that layout (25 / 104 / 274 wavefronts on scale
    17 / 20 / 22): a calibration on ONE launch, made with the last GPU minutes of the round -- the 104 wavefronts were set through GEMHIP_SGNS_MAX_WAVES,
    which gives the same launch the rule now plans (tests/test_sgns_plan.py); scale 17 at 25 wavefronts has not been run."""
    import json, os
    from conftest import golden_path
    from gem_amd.evaluation import reconstruction as gr
    # scale 17: 16 384 eligible nodes (their APs sum to 288: s.e. 0.3 %); scale 20: 131 072 (sum 384; over 16 384 the APs sum to 49 and one launch has an s.e. of 1.4 %)
    path = golden_path('n2v_ref_oracle_rmat%d%s_%s.json' % (scale, '' if layout == 'node_id' else '_vocab_order', 'e16k' if scale == 17 else 'e128k'))
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
    bar = 0.03 if scale == 17 else 0.05          # (scale 20: measured -1.51 % (s.e. 0.34 %) / -1.50 % (s.e. 0.64 %) in the binary's / the node-id layout: >= 5 s.e. of margin)
    record_stat('R-MAT scale %d, %s layout, one Hogwild launch against the sequential oracle (paired, %d nodes)' % (scale, layout, len(d)),
                '%+.2f %% (s.e. %.2f %%)' % (100 * gap, 100 * se), '+-%d %%' % round(100 * bar))
    assert abs(gap) <= bar, (gap, se, ap.mean(), ref['MAP'])
