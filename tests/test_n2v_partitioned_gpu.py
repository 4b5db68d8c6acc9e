"""GPU tests of the partitioned (multi-GPU episode) SGNS path -- gemhip_sgns_train_part, the walk-ordered bucket kernel -- and of the 16k-node parity point."""
import ctypes as C
import json

import numpy as np
import pytest
import torch

import oracle
from gem_amd import _hip, multi_gpu
from gem_amd.embedding.node2vec import node2vec
from gem_amd.evaluation import reconstruction as gr
from gem_amd.graph import edge_arrays, sbm_graph, to_csr
from conftest import golden_path

pytestmark = pytest.mark.gpu


def backend(G, d):
    n, src, dst, w, _ = edge_arrays(G)
    row_ptr, col, ww = to_csr(n, src, dst, w)
    return n, src, dst, multi_gpu.HipBackendN2V(n, row_ptr, col, ww, d)


def _walks_of(b, l):
    nw = C.c_int64(); wl = C.c_int32(); p = C.c_void_p()
    _hip.check(b.L.gemhip_n2v_walks_ptr(b.h, C.byref(p), C.byref(nw), C.byref(wl)))
    walks = np.empty((nw.value, l), np.int32)
    _hip.check(b.L.gemhip_n2v_get_walks(b.h, _hip.ptr(walks, C.c_int32)))
    return walks


@pytest.mark.parametrize('d,parts,bucket,flags,hogwild_path,segments', [(16, 4, (1, 0), 9, 0, False), (128, 2, (0, 1), 11, 0, False), (7, 3, (2, 2), 9, 0, True),
                                                                      (128, 2, (1, 1), 11, 1, True), (64, 5, (3, 1), 11, 9, False), (129, 2, (0, 1), 11, 1, False)])
def test_train_part_deterministic_matches_oracle(sbm1024, d, parts, bucket, flags, hogwild_path, segments):
    """gemhip_sgns_train_part (sgns_win_kernel<PART>) on ONE wavefront == TrainModel in walk order restricted to the bucket (context partition,
    word partition) -- oracle_sgns_train_part, the restatement of ELF @0x40d6a0 with the partition filter -- to the 2e-4 of the unpartitioned
    test.  hogwild_path 1 runs the Hogwild instantiation (delta write-back, reload-on-update, atomic centre row) on the one wavefront, 9 the same
    with every node of >= 30 tokens hot (never in the LDS window, negatives by atomic add).  segments: the corpus as two shards of unequal
    length with their own walk ids (the N-rank layout), including a padded tail."""
    n, src, dst, b = backend(sbm1024, d)
    l = 40
    b.walks(1.0, 1.0, 1, l, 3, flags, 100, 356)
    walks = _walks_of(b, l)
    b.vocab(); b.build_unigram_parts(parts)
    counts = b.counts.cpu().numpy()
    UT, KT, off = oracle.unigram_build_parts(counts, parts)
    UTd = np.empty(n, np.float32); KTd = np.empty(n, np.int32)
    _hip.check(b.L.gemhip_n2v_build_unigram_parts(b.h, parts, _hip.ptr(UTd, C.c_float), _hip.ptr(KTd, C.c_int32)))
    assert np.array_equal(UTd, UT) and np.array_equal(KTd, KT)
    gi, gj = bucket
    P, _, _ = b.init_part_tables(3, gi, parts)
    rows = P.shape[0]
    Np = (0.05 * torch.randn((rows, d), generator=torch.Generator().manual_seed(1))).to(P.device)
    Po, No = P.cpu().numpy().copy(), Np.cpu().numpy().copy()
    if hogwild_path:
        _hip.check(b.L.gemhip_sgns_set_window_cache(b.h, -1, 1))
        if hogwild_path == 9:
            _hip.check(b.L.gemhip_sgns_set_hot_rows(b.h, 30))
    dev = P.device
    if segments:      # shard 0 = walks [0, 150) with ids 100.., shard 1 = walks [150, 256) with ids 250..; work items 2 x 150 (44 of them absent)
        corpus = torch.from_numpy(walks).to(dev)
        tab = torch.tensor([[[0, 150], [150, 106], [100, 250]]], dtype=torch.int64, device=dev)
        b.train_part(corpus, tab, 0, 2, 150, 5, 0.025, 2 * 150 * l, 1000, 0, 3, flags | 4, gi, gj, P, Np)
        want_pairs = 0
        for r, (lo, cnt, wid0) in enumerate(((0, 150, 100), (150, 106, 250))):
            want_pairs += oracle.sgns_train_part(walks[lo:lo + cnt], None, 5, 0.025, 2 * 150 * l, 1000 + r * 150 * l, 0, parts, gi, gj, UT[off[gj]:off[gj + 1]],
                                                 KT[off[gj]:off[gj + 1]], 3, flags, Po, No, walk_id_offset=wid0, local_rows=True)
    else:
        corpus = torch.from_numpy(walks).to(dev)
        _hip.check(b.L.gemhip_sgns_train_part(b.h, C.c_void_p(corpus.data_ptr()), 256, l, None, 0, 0, 100, 5, 0.025, 256 * l, 0, 0, 3, flags | 4, gi, gj,
                                              C.c_void_p(P.data_ptr()), C.c_void_p(Np.data_ptr()), d, None))
        want_pairs = oracle.sgns_train_part(walks, None, 5, 0.025, 256 * l, 0, 0, parts, gi, gj, UT[off[gj]:off[gj + 1]], KT[off[gj]:off[gj + 1]], 3, flags,
                                            Po, No, walk_id_offset=100, local_rows=True)
    torch.cuda.synchronize()
    assert b.pairs() == want_pairs > 100
    for got, want in ((P.cpu().numpy(), Po), (Np.cpu().numpy(), No)):
        assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max() + 1e-6
    b.close()


def test_train_part_with_one_partition_is_train(sbm1024):
    """parts == 1: the bucket is everything and gemhip_sgns_train_part is gemhip_sgns_train -- same kernel body, same draws; the two differ only in
    WHICH pairs take the exact sequential path (the bucket kernel tests a slot against the previous two slots that have a context, the plain kernel
    against the previous two slots), i.e. in the summation order of a few dot products: equal to the 2e-4 of the oracle tests."""
    n, src, dst, b = backend(sbm1024, 32)
    b.walks(1.0, 1.0, 1, 40, 5, 11, 0, 200)
    b.vocab(); b.build_unigram(); b.build_unigram_parts(1)
    P, N = b.init_tables(5)
    b.train(10, 1, 0, 0, 200, 200 * 40, 0, 5, 11 | 4)
    torch.cuda.synchronize()
    P1, N1 = P.clone(), N.clone()
    b.init_tables(5)
    nw = C.c_int64(); wl = C.c_int32(); p = C.c_void_p()
    _hip.check(b.L.gemhip_n2v_walks_ptr(b.h, C.byref(p), C.byref(nw), C.byref(wl)))
    _hip.check(b.L.gemhip_sgns_train_part(b.h, p, 200, 40, None, 0, 0, 0, 10, 0.025, 200 * 40, 0, 0, 5, 11 | 4, 0, 0, C.c_void_p(P.data_ptr()),
                                          C.c_void_p(N.data_ptr()), 32, None))
    torch.cuda.synchronize()
    for a, z in ((P, P1), (N, N1)):
        assert float((a - z).abs().max()) <= 2e-4 * float(z.abs().max()) + 1e-6
    b.close()


@pytest.mark.hogwild_stat
@pytest.mark.parametrize('parts', [1, 4])
def test_partitioned_schedule_quality_on_one_gpu(sbm1024, parts):
    """The episode schedule through the HIP backend with `parts` virtual ranks (Node2VecPartitioned.run_virtual; parts = 1 is the real world-1 run):
    every pair of TrainModel is trained exactly once and the MAP equals the sequential algorithm's.  On a 1024-node graph Hogwild itself costs a
    few percent (the launch rule bounds the open fraction of a 1024 / parts-row partition) -- hence the 8 % bar over 3 seeds."""
    n, src, dst, b = backend(sbm1024, 16)
    maps = []
    rp, col, _ = oracle.sorted_csr(n, src, dst, None)
    for seed in (1, 2, 3):
        job = multi_gpu.Node2VecPartitioned(b, multi_gpu.TorchComm(1), 0, 1, n, 10, 80, 10, 1, seed=seed, flags=9, episodes=16)
        P = (job.run(1.0, 1.0) if parts == 1 else job.run_virtual(parts)).cpu().numpy().astype(np.float64)
        if seed == 1:
            assert job.pairs_trained == len(oracle.sgns_pairs(oracle.n2v_walks(rp, col, None, None, 1.0, 1.0, 10, 80, 1, 9), 10, 0, 0, 1))
        m = node2vec(d=16, max_iter=1, walk_len=80, num_walks=10, con_size=10, ret_p=1, inout_p=1)
        maps.append(gr.evaluateStaticGraphReconstruction(sbm1024, m, P, None)[0])
    Xs, _ = oracle.n2v_train(n, src, dst, None, 16, 80, 10, 10, 1, 1.0, 1.0, 1, 9)
    MAPs = gr.evaluateStaticGraphReconstruction(sbm1024, m, Xs.astype(np.float64), None)[0]
    assert abs(np.mean(maps) - MAPs) <= 0.08 * MAPs, (maps, MAPs)
    b.close()


@pytest.mark.hogwild_stat
def test_partitioned_16k_four_virtual_ranks_match_race_free_snap():
    """SBM 16384 / d = 128 with FOUR virtual ranks (16 buckets per episode, 64 episodes): the MAP of the partitioned schedule equals the reference
    binary's race-free run (tests/golden/n2v_ref_16k.json: 0.926) within 1 % -- the N > 1 counterpart of the single-GPU test below."""
    ref = json.load(open(golden_path('n2v_ref_16k.json')))
    p = ref['params']
    g = sbm_graph(p['n'], p['edges'], p['blocks'], p['seed'])
    n, src, dst, b = backend(g, p['d'])
    job = multi_gpu.Node2VecPartitioned(b, multi_gpu.TorchComm(1), 0, 1, n, p['num_walks'], p['walk_len'], p['window'], 1, seed=3, flags=11, episodes=64)
    P = job.run_virtual(4).cpu().numpy().astype(np.float64)
    m = node2vec(d=p['d'], max_iter=1, walk_len=p['walk_len'], num_walks=p['num_walks'], con_size=p['window'], ret_p=1, inout_p=1)
    MAP = gr.evaluateStaticGraphReconstruction(g, m, P, None)[0]
    assert abs(MAP - ref['snap']['t1']['MAP']) <= 0.01 * ref['snap']['t1']['MAP'], (MAP, ref['snap'])
    b.close()


@pytest.mark.hogwild_stat
def test_sbm16k_map_matches_race_free_snap():
    """SBM 16384 nodes / 164k edges, d=128: the real binary single-threaded reaches MAP 0.926 (8 threads: 0.289,
    tests/golden/n2v_ref_16k.json, scripts/make_golden_n2v_16k.py).  The HIP path (Hogwild, auto width) must match the
    race-free value within 1 %."""
    ref = json.load(open(golden_path('n2v_ref_16k.json')))
    p = ref['params']
    g = sbm_graph(p['n'], p['edges'], p['blocks'], p['seed'])
    m = node2vec(d=p['d'], max_iter=1, walk_len=p['walk_len'], num_walks=p['num_walks'], con_size=p['window'], ret_p=1, inout_p=1, seed=3)
    Y = m.learn_embedding(graph=g, is_weighted=True, no_python=True)
    MAP = gr.evaluateStaticGraphReconstruction(g, m, Y, None)[0]
    assert abs(MAP - ref['snap']['t1']['MAP']) <= 0.01 * ref['snap']['t1']['MAP'], (MAP, ref['snap'])
    assert MAP > ref['snap']['t8']['MAP']


@pytest.mark.hogwild_stat
def test_partitioned_1m_two_virtual_ranks_inside_the_stated_n_gpu_tolerance():
    """BASELINE configs[3] at N = 2: the partitioned schedule on SBM 1M/10M with two VIRTUAL ranks (the one-GPU box; the buckets of a round touch
    disjoint rows, so running them rank after rank computes what two GPUs compute), 64 episodes, the plugin's table layout, against the sequential oracle's
    run on the same seed (tests/golden/n2v_ref_oracle_1000k_vocab_order_s4096.json, paired over its 4 096 nodes).  The schedule trains a walk's pairs in N^2
    passes spread over an episode instead of back to back, which optimises slightly BETTER than TrainModel's order (+2.8 ... +3.1 % at N = 2, 4, 8 in round 4;
    the schedule's CPU emulation without any concurrency: +0.7 ... +1.7 %): the stated N >= 2 tolerance is +4 / -1 %, asserted here."""
    import os
    path = golden_path('n2v_ref_oracle_1000k_vocab_order_s4096.json')
    if not os.path.exists(path):
        pytest.skip('no 1M oracle run in the plugin layout committed')
    ref = json.load(open(path))
    p = ref['params']
    g = sbm_graph(p['n'], p['edges'], p['blocks'], p['seed'])
    n, src, dst, b = backend(g, p['d'])
    b.vocab_order = True
    job = multi_gpu.Node2VecPartitioned(b, multi_gpu.TorchComm(1), 0, 1, n, p['num_walks'], p['walk_len'], p['window'], 1, seed=20260923, flags=p['flags'], episodes=64)
    P = job.run_virtual(2).cpu().numpy()
    nodes = np.random.RandomState(0).choice(g.n, size=len(ref['ap']), replace=False)
    ap = gr.sampled_ap_gpu(g, None, P, nodes)
    dd = ap - np.asarray(ref['ap'])
    gap, se = float(dd.mean() / ref['MAP']), float(dd.std(ddof=1) / np.sqrt(len(dd)) / ref['MAP'])
    from conftest import record_stat
    record_stat('SBM 1M/10M, partitioned schedule with 2 virtual ranks against the sequential oracle (paired, %d nodes)' % len(dd),
                '%+.2f %% (s.e. %.2f %%); kernel seconds per rank %s' % (100 * gap, 100 * se, [round(v, 2) for v in getattr(job, 'virtual_rank_seconds', [])]), '-1 ... +4 %')
    assert -0.01 <= gap <= 0.04, (gap, se)
    b.close()


def test_locally_hot_rows_reach_the_bucket_launches():
    """Round 6: the partitioned schedule builds the locally-hot key from the GATHERED corpus (gemhip_n2v_locally_hot_corpus, called by Node2VecPartitioned and by
    gemhip_n2v_train_multi): on an SBM with a dozen isolated edges appended exactly their ends qualify, a Hogwild bucket launch then carries the hot-row
    machinery with the threshold INT32_MAX (only those nodes), and the deterministic schedule is untouched (equal to the run with the rule off, bit for bit)."""
    from test_n2v_gpu import _graph_with_isolated_edges
    n, src, dst = _graph_with_isolated_edges(n_core=16384, iso_pairs=12)
    row_ptr, col, ww = to_csr(n, src, dst, None)
    res = {}
    for rule in (8, 0):
        b = multi_gpu.HipBackendN2V(n, row_ptr, col, ww, 32)
        job = multi_gpu.Node2VecPartitioned(b, multi_gpu.TorchComm(1), 0, 1, n, 2, 40, 5, 1, seed=3, flags=9, episodes=2)
        if rule == 0:
            b.locally_hot = lambda corpus: 0                 # rule off: the driver's hook does nothing
        P = job.run_virtual(2)
        k, wv, hot, fr = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        _hip.check(b.L.gemhip_sgns_last_launch(b.h, C.byref(k), C.byref(wv), C.byref(hot), C.byref(fr)))
        res[rule] = (hot.value, wv.value, bool(torch.isfinite(P).all()))
        b.close()
    assert res[8][0] == np.iinfo(np.int32).max and res[8][1] > 1 and res[8][2]
    assert res[0][0] == 0 and res[0][2]
    # the count itself, and the deterministic schedule with and without the rule
    b = multi_gpu.HipBackendN2V(n, row_ptr, col, ww, 32)
    job = multi_gpu.Node2VecPartitioned(b, multi_gpu.TorchComm(1), 0, 1, n, 2, 40, 5, 1, seed=3, flags=9 | 4, episodes=2)
    Pa = job.run_virtual(2).cpu().numpy()
    corpus = b.gather_corpus(multi_gpu.TorchComm(1), job.shard_rows, 1)
    assert b.locally_hot(corpus) == 12 * 2 + 3
    b.locally_hot = lambda corpus: 0
    Pb = multi_gpu.Node2VecPartitioned(b, multi_gpu.TorchComm(1), 0, 1, n, 2, 40, 5, 1, seed=3, flags=9 | 4, episodes=2).run_virtual(2).cpu().numpy()
    assert np.array_equal(Pa, Pb)
    b.close()
