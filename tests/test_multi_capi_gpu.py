"""GPU tier: the N-GPU entry points of the C ABI (gem_amd/csrc/multi.hip: gemhip_gf_train_multi, gemhip_n2v_train_multi, gemhip_rccl_selftest).
The test box has ONE GPU: n_gpus = 1 runs the real RCCL communicator (1-rank collectives), n_gpus > 1 runs VIRTUAL ranks on device 0 (a device list
that repeats the device: collectives become copies, sharding / schedule / kernels are the production code)."""
import ctypes as C

import numpy as np
import pytest

import oracle
from gem_amd import _hip
from gem_amd.embedding.node2vec import node2vec
from gem_amd.evaluation import reconstruction as gr
from gem_amd.graph import edge_arrays, sbm_graph, to_csr

pytestmark = pytest.mark.gpu


def _devs(k):
    return None if k == 1 else (C.c_int32 * k)(*([0] * k))


def test_rccl_selftest_one_rank_and_virtual_ranks():
    """Communicator create -> all-gather / all-reduce / ring shift -> destroy.  One rank goes through RCCL itself (ncclCommInitAll on one device; a
    1-rank all-gather is a copy, the ring shift a self send/recv); 2 and 4 virtual ranks check the rank arithmetic of the three collectives."""
    L = _hip.lib()
    sec = C.c_double()
    _hip.check(L.gemhip_rccl_selftest(1, None, 1 << 20, C.byref(sec)))
    assert sec.value > 0
    for k in (2, 4):
        _hip.check(L.gemhip_rccl_selftest(k, _devs(k), 4096, None))
    assert L.gemhip_rccl_selftest(2, None, 4096, None) != 0            # two REAL ranks need two GPUs: an error, never a silent 1-GPU run
    assert b'device 1 of 1' in L.gemhip_last_error()
    assert L.gemhip_rccl_selftest(0, None, 4096, None) != 0


@pytest.mark.parametrize('ranks', [1, 2, 4])
def test_gf_train_multi_is_bit_identical_to_one_gpu(ranks):
    """Source rows in `ranks` contiguous blocks, all-gather of the owned blocks after every sweep == gemhip_gf_train, bit for bit (n not divisible by
    the rank count: padded blocks); and == the oracle to the usual tolerance."""
    g = sbm_graph(1001, 10000, 4, seed=3)
    n, src, dst, w, _ = edge_arrays(g)
    X0 = (0.05 * np.random.RandomState(0).randn(n, 32)).astype(np.float32)
    L = _hip.lib()
    a = X0.copy(); b = X0.copy()
    _hip.check(L.gemhip_gf_train(n, len(src), _hip.ptr(src, C.c_int32), _hip.ptr(dst, C.c_int32), None, 32, 0.05, 0.01, 6, _hip.ptr(a, C.c_float), None))
    st = (C.c_double * 8)()
    _hip.check(L.gemhip_gf_train_multi(n, len(src), _hip.ptr(src, C.c_int32), _hip.ptr(dst, C.c_int32), None, 32, 0.05, 0.01, 6, ranks, _devs(ranks),
                                       _hip.ptr(b, C.c_float), st))
    assert np.array_equal(a, b)
    assert st[4] == ranks and st[5] == (1.0 if ranks > 1 else 0.0) and st[1] > 0
    ref = oracle.gf_train_f32(n, src, dst, None, 32, 0.05, 0.01, 6, X0)
    assert np.abs(b - ref).max() <= 2e-5 * np.abs(ref).max()


def test_gf_train_multi_refuses_an_order_it_cannot_shard(karate):
    """karate's insertion order 0, 31, 21, ... visits sources out of id order: sharded ranks would read rows 'already updated' in the same sweep.
    One GPU handles it (levels); n_gpus > 1 says so instead of training something else."""
    n, src, dst, w, _ = edge_arrays(karate)
    X = (0.1 * np.random.RandomState(1).randn(n, 8)).astype(np.float32)
    L = _hip.lib()
    rc = L.gemhip_gf_train_multi(n, len(src), _hip.ptr(src, C.c_int32), _hip.ptr(dst, C.c_int32), None, 8, 0.05, 0.01, 3, 2, _devs(2), _hip.ptr(X.copy(), C.c_float), None)
    assert rc == -3 and b'ascending id order' in L.gemhip_last_error()
    Y = X.copy()
    _hip.check(L.gemhip_gf_train_multi(n, len(src), _hip.ptr(src, C.c_int32), _hip.ptr(dst, C.c_int32), None, 8, 0.05, 0.01, 3, 1, None, _hip.ptr(Y, C.c_float), None))
    assert np.abs(Y - oracle.gf_train_f32(n, src, dst, None, 8, 0.05, 0.01, 3, X)).max() <= 2e-5 * np.abs(Y).max()


def _oracle_multi_schedule(n, src, dst, d, L, r, win, seed, flags, N, episodes):
    """gemhip_n2v_train_multi restated with the oracle: the same shards, episode slices, bucket order (episode, round s, rank g -> bucket (g, (g+s) % N)),
    alpha offsets and partition tables, every bucket through oracle_sgns_train_part."""
    rp, cs, _ = oracle.sorted_csr(n, src, dst, None)
    total = len(oracle.start_nodes(rp, cs)) * r
    shard = [(total * k // N, total * (k + 1) // N) for k in range(N)]
    walks = [oracle.n2v_walks(rp, cs, None, None, 1.0, 1.0, r, L, seed, flags, lo, hi) for lo, hi in shard]
    cnt = sum(oracle.n2v_vocab(n, w_) for w_ in walks)
    if flags & _hip.N2V_VOCAB_ORDER:      # the binary's layout per partition: first appearance in the corpus of all ranks (walk-id order)
        tabs = oracle.unigram_build_parts_vocab_order(cnt.astype(np.int32), np.concatenate(walks, axis=0), N, flags)
    else:
        UTp, KTp, off = oracle.unigram_build_parts(cnt.astype(np.int32), N)
        tabs = [(None, UTp[off[k]:off[k + 1]], KTp[off[k]:off[k + 1]]) for k in range(N)]
    P, Nn = oracle.sgns_init(n, d, seed)
    Pl = [np.ascontiguousarray(P[k::N]) for k in range(N)]
    Nl = [np.zeros_like(Pl[k]) for k in range(N)]
    seg_len = [max(1, max((hi - lo) * (e + 1) // episodes - (hi - lo) * e // episodes for lo, hi in shard)) for e in range(episodes)]
    alpha_total = sum(s_ * N * L for s_ in seg_len)
    done, pairs = 0, 0
    for e in range(episodes):
        for s in range(N):
            for g in range(N):
                h = (g + s) % N
                for k, (lo, hi) in enumerate(shard):          # work items: shard after shard
                    a, z = (hi - lo) * e // episodes, (hi - lo) * (e + 1) // episodes
                    if z > a:
                        pairs += oracle.sgns_train_part(walks[k][a:z], None, win, 0.025, alpha_total, done + k * seg_len[e] * L, 0, N, g, h,
                                                        tabs[h][1], tabs[h][2], seed, flags, Pl[g], Nl[h], walk_id_offset=lo + a,
                                                        local_rows=True, slot_tab=tabs[h][0])
        done += seg_len[e] * N * L
    X = np.zeros((n, d), np.float32)
    for k in range(N):
        X[k::N] = Pl[k]
    return X, pairs


@pytest.mark.parametrize('ranks,episodes', [(1, 3), (3, 4)])
def test_n2v_train_multi_deterministic_equals_the_oracle_schedule(sbm1024, ranks, episodes):
    """flags | 4 (every bucket on ONE wavefront in walk order): the C-ABI driver -- shards, count all-reduce, corpus all-gather, episode table, bucket
    order, ring of SynNeg partitions, assembly -- lands on the oracle's restatement of the same schedule to 2e-4, and trains every pair of TrainModel
    exactly once."""
    n, src, dst, w, _ = edge_arrays(sbm1024)
    row_ptr, col, _ = to_csr(n, src, dst, None)
    d, L, r, win, seed, flags = 16, 30, 1, 5, 7, 11
    X = np.empty((n, d), np.float32)
    st = (C.c_double * 8)()
    _hip.check(_hip.lib().gemhip_n2v_train_multi(n, len(col), _hip.ptr(row_ptr, C.c_int64), _hip.ptr(col, C.c_int32), None, d, L, r, win, 1, 1.0, 1.0, seed,
                                                 flags | 4, ranks, _devs(ranks), episodes, _hip.ptr(X, C.c_float), st))
    want, pairs = _oracle_multi_schedule(n, src, dst, d, L, r, win, seed, flags, ranks, episodes)
    assert st[3] == pairs and st[5] == ranks
    rp, cs, _ = oracle.sorted_csr(n, src, dst, None)
    assert pairs == len(oracle.sgns_pairs(oracle.n2v_walks(rp, cs, None, None, 1.0, 1.0, r, L, seed, flags), win, 0, 0, seed))
    assert np.abs(X - want).max() <= 2e-4 * np.abs(want).max() + 1e-6


@pytest.mark.hogwild_stat
@pytest.mark.parametrize('ranks', [1, 4])
def test_n2v_train_multi_quality(sbm1024, ranks):
    """Hogwild buckets: graph-reconstruction MAP of the N-rank run against the sequential algorithm (same bar as the partitioned driver of
    gem_amd/multi_gpu.py on this 1024-node graph: 8 % over 3 seeds)."""
    n, src, dst, w, _ = edge_arrays(sbm1024)
    row_ptr, col, _ = to_csr(n, src, dst, None)
    m = node2vec(d=16, max_iter=1, walk_len=80, num_walks=10, con_size=10, ret_p=1, inout_p=1)
    maps = []
    for seed in (1, 2, 3):
        X = np.empty((n, 16), np.float32)
        _hip.check(_hip.lib().gemhip_n2v_train_multi(n, len(col), _hip.ptr(row_ptr, C.c_int64), _hip.ptr(col, C.c_int32), None, 16, 80, 10, 10, 1, 1.0, 1.0, seed, 9,
                                                     ranks, _devs(ranks), 16, _hip.ptr(X, C.c_float), None))
        maps.append(gr.evaluateStaticGraphReconstruction(sbm1024, m, X.astype(np.float64), None)[0])
    Xs, _ = oracle.n2v_train(n, src, dst, None, 16, 80, 10, 10, 1, 1.0, 1.0, 1, 9)
    MAPs = gr.evaluateStaticGraphReconstruction(sbm1024, m, Xs.astype(np.float64), None)[0]
    assert abs(np.mean(maps) - MAPs) <= 0.08 * MAPs, (maps, MAPs)


def test_multi_entry_points_reject_bad_rank_counts_before_sizing_anything():
    """n_gpus <= 0 or > 64: GEMHIP_E_INVALID from all three entry points (round 4 divided by zero at n_gpus = 0 and threw across the C ABI at -1)."""
    g = sbm_graph(300, 3000, 2, seed=3)
    n, src, dst, w, _ = edge_arrays(g)
    row_ptr, col, _ = to_csr(n, src, dst, None)
    L = _hip.lib()
    X = np.zeros((n, 8), np.float32)
    for bad in (0, -1, -1000, 65):
        assert L.gemhip_rccl_selftest(bad, None, 4096, None) == _hip.E_INVALID
        assert L.gemhip_gf_train_multi(n, len(src), _hip.ptr(src, C.c_int32), _hip.ptr(dst, C.c_int32), None, 8, 0.05, 0.01, 1, bad, None, _hip.ptr(X, C.c_float),
                                       None) == _hip.E_INVALID
        assert L.gemhip_n2v_train_multi(n, len(col), _hip.ptr(row_ptr, C.c_int64), _hip.ptr(col, C.c_int32), None, 8, 20, 1, 5, 1, 1.0, 1.0, 1, 11, bad, None, 2,
                                        _hip.ptr(X, C.c_float), None) == _hip.E_INVALID
        assert b'n_gpus' in L.gemhip_last_error()


@pytest.mark.parametrize('ranks,episodes', [(1, 2), (3, 4)])
def test_n2v_train_multi_in_the_binarys_table_layout(sbm1024, ranks, episodes):
    """flags with GEMHIP_N2V_VOCAB_ORDER (27, the plugin default): the partition tables are laid out over each partition's nodes in order of first
    appearance in the whole corpus (round 4 ignored the bit and trained with node-id-order tables: ADVICE r4).  Deterministic mode: one device equals
    gemhip_n2v_train bit for bit (one partition IS the binary's table), and 1 / 3 virtual ranks land on the oracle's restatement of the schedule with
    oracle.unigram_build_parts_vocab_order's tables (2e-4, every pair trained once) -- and NOT on the node-id-order tables' result."""
    n, src, dst, w, _ = edge_arrays(sbm1024)
    row_ptr, col, _ = to_csr(n, src, dst, None)
    L = _hip.lib()
    d, Lw, r, win, seed, flags = 16, 30, 2, 5, 7, _hip.N2V_SNAP_LAYOUT
    X = np.empty((n, d), np.float32)
    st = (C.c_double * 8)()
    args = (n, len(col), _hip.ptr(row_ptr, C.c_int64), _hip.ptr(col, C.c_int32), None, d, Lw, r, win, 1, 1.0, 1.0, seed)
    _hip.check(L.gemhip_n2v_train_multi(*args, flags | 4, ranks, _devs(ranks), episodes, _hip.ptr(X, C.c_float), st))
    want, pairs = _oracle_multi_schedule(n, src, dst, d, Lw, r, win, seed, flags, ranks, episodes)
    assert st[3] == pairs and st[5] == ranks
    assert np.abs(X - want).max() <= 2e-4 * np.abs(want).max() + 1e-6
    other, _ = _oracle_multi_schedule(n, src, dst, d, Lw, r, win, seed, flags & ~_hip.N2V_VOCAB_ORDER, ranks, episodes)
    assert np.abs(X - other).max() > 1e-2 * np.abs(want).max()                # the layout is not cosmetic: other negative draws
    if ranks == 1:
        a = np.empty((n, d), np.float32)
        _hip.check(L.gemhip_n2v_train(*args, flags | 4, _hip.ptr(a, C.c_float), None))
        assert np.abs(a - X).max() <= 2e-4 * np.abs(a).max()               # (the same pass cut into episode launches)


def test_build_unigram_parts_vocab_order_tables(sbm1024):
    """gemhip_n2v_build_unigram_parts_vocab_order against oracle.unigram_build_parts_vocab_order on the handle's own walks: slot tables, slot counts and the
    alias arrays by local row, for 1 and 4 partitions and both settings of RndUnigramInt's quirk bit; one partition == the single-table builder."""
    from test_n2v_gpu import Dev
    n, src, dst, w, _ = edge_arrays(sbm1024)
    dev = Dev(n, src, dst, w)
    walks = dev.walks(1.0, 1.0, 2, 30, 5, 11)
    cnt, _, _ = dev.unigram()
    for parts in (1, 4):
        for flags in (27, 27 & ~2):
            UT = np.empty(n, np.float32); KT = np.empty(n, np.int32); SL = np.empty(n, np.int32); ns = np.empty(parts, np.int64)
            _hip.check(dev.L.gemhip_n2v_build_unigram_parts_vocab_order(dev.h, parts, flags, None, 0, _hip.ptr(UT, C.c_float), _hip.ptr(KT, C.c_int32),
                                                                        _hip.ptr(SL, C.c_int32), _hip.ptr(ns, C.c_int64)))
            tabs = oracle.unigram_build_parts_vocab_order(cnt, walks, parts, flags)
            off = 0
            for p, (slot, U, K) in enumerate(tabs):
                npart = len(U)
                assert ns[p] == len(slot) and np.array_equal(SL[off:off + len(slot)], slot) and np.all(SL[off + len(slot):off + npart] == -1)
                assert np.array_equal(UT[off:off + npart], U) and np.array_equal(KT[off:off + npart], K)
                off += npart
    dev.close()


def _real_devices(k):
    import torch
    if torch.cuda.device_count() < k:
        pytest.skip('needs %d GPUs on one node (the test pool has one): the RCCL path across DISTINCT devices stays unvalidated until this runs' % k)
    return (C.c_int32 * k)(*range(k))


@pytest.mark.parametrize('ranks', [2, 4])
def test_real_devices_gf_train_multi_is_bit_identical_to_one_gpu(ranks):
    """The assertions of the virtual-rank test on DISTINCT devices: grouped in-place ncclAllGather over RCCL/xGMI after every sweep."""
    devs = _real_devices(ranks)
    g = sbm_graph(20011, 200000, 8, seed=3)
    n, src, dst, w, _ = edge_arrays(g)
    X0 = (0.05 * np.random.RandomState(0).randn(n, 128)).astype(np.float32)
    L = _hip.lib()
    a = X0.copy(); b = X0.copy()
    _hip.check(L.gemhip_gf_train(n, len(src), _hip.ptr(src, C.c_int32), _hip.ptr(dst, C.c_int32), None, 128, 0.05, 0.01, 12, _hip.ptr(a, C.c_float), None))
    st = (C.c_double * 8)()
    _hip.check(L.gemhip_gf_train_multi(n, len(src), _hip.ptr(src, C.c_int32), _hip.ptr(dst, C.c_int32), None, 128, 0.05, 0.01, 12, ranks, devs, _hip.ptr(b, C.c_float), st))
    assert np.array_equal(a, b) and st[4] == ranks and st[5] == 0.0
    _hip.check(L.gemhip_rccl_selftest(ranks, devs, 1 << 22, None))


@pytest.mark.parametrize('ranks,episodes', [(2, 3), (4, 4)])
def test_real_devices_n2v_train_multi_deterministic_equals_the_oracle_schedule(sbm1024, ranks, episodes):
    """The deterministic schedule test on DISTINCT devices: count all-reduce, corpus all-gather and the ncclSend / ncclRecv ring of SynNeg partitions."""
    devs = _real_devices(ranks)
    n, src, dst, w, _ = edge_arrays(sbm1024)
    row_ptr, col, _ = to_csr(n, src, dst, None)
    d, Lw, r, win, seed, flags = 16, 30, 1, 5, 7, 11
    X = np.empty((n, d), np.float32)
    st = (C.c_double * 8)()
    _hip.check(_hip.lib().gemhip_n2v_train_multi(n, len(col), _hip.ptr(row_ptr, C.c_int64), _hip.ptr(col, C.c_int32), None, d, Lw, r, win, 1, 1.0, 1.0, seed,
                                                 flags | 4, ranks, devs, episodes, _hip.ptr(X, C.c_float), st))
    want, pairs = _oracle_multi_schedule(n, src, dst, d, Lw, r, win, seed, flags, ranks, episodes)
    assert st[3] == pairs and st[5] == ranks and st[6] == 0.0
    assert np.abs(X - want).max() <= 2e-4 * np.abs(want).max() + 1e-6


# ----------------------------------------------------------------------------------------------- n_gpus through the plugin API (round 6)
@pytest.mark.parametrize('ranks', [2, 4])
def test_plugin_gf_n_gpus_is_the_one_gpu_table(ranks):
    """GraphFactorization(n_gpus=N, virtual_ranks=True).learn_embedding() -> gemhip_gf_train_multi: the same table as the one-GPU plugin call with the
    same numpy draw (device_init=False), bit for bit; the stats say which driver ran."""
    from gem_amd.embedding.gf import GraphFactorization
    g = sbm_graph(1001, 10000, 4, seed=3)
    kw = dict(d=32, eta=0.05, regu=0.01, max_iter=6, seed=5)
    one = GraphFactorization(device_init=False, **kw)
    X1 = one.learn_embedding(graph=g, is_weighted=True, no_python=True)
    many = GraphFactorization(n_gpus=ranks, virtual_ranks=True, **kw)
    XN = many.learn_embedding(graph=g, is_weighted=True, no_python=True)
    assert XN.dtype == np.float64 and np.array_equal(X1, XN)
    # (seed=s with device_init=False is the numpy draw RandomState(s) through the one-shot gemhip_gf_train path -- the reference convention's
    # reproducible form; seed=s alone draws on the device since round 5)
    n, src, dst, w, _ = edge_arrays(g)
    X0 = (0.01 * np.random.RandomState(5).randn(n, 32)).astype(np.float32)
    ref = oracle.gf_train_f32(n, src, dst, None, 32, 0.05, 0.01, 6, X0)
    assert np.abs(X1 - ref).max() <= 2e-5 * np.abs(ref).max()
    assert many._stats['driver'] == 'gemhip_gf_train_multi' and many._stats['n_gpus'] == ranks and many._stats['virtual_ranks']
    assert 'n_gpus' not in GraphFactorization.hyper_params


def test_plugin_node2vec_n_gpus_deterministic_is_the_capi_call(sbm1024):
    """node2vec(n_gpus=3, virtual_ranks=True, episodes=4, flags=... | 4): the plugin hands exactly its hyper-parameters to gemhip_n2v_train_multi
    (same embedding as the direct C-ABI call, which tests above tie to the oracle's schedule)."""
    n, src, dst, w, _ = edge_arrays(sbm1024)
    row_ptr, col, _ = to_csr(n, src, dst, None)
    d, L, r, win, seed, flags = 16, 30, 1, 5, 7, 11 | 4
    m = node2vec(d=d, max_iter=1, walk_len=L, num_walks=r, con_size=win, ret_p=1, inout_p=1, seed=seed, flags=flags, n_gpus=3, virtual_ranks=True, episodes=4)
    X = m.learn_embedding(graph=sbm1024, is_weighted=True, no_python=True)
    want = np.empty((n, d), np.float32)
    _hip.check(_hip.lib().gemhip_n2v_train_multi(n, len(col), _hip.ptr(row_ptr, C.c_int64), _hip.ptr(col, C.c_int32), None, d, L, r, win, 1, 1.0, 1.0, seed,
                                                 flags, 3, _devs(3), 4, _hip.ptr(want, C.c_float), None))
    assert np.array_equal(X, want.astype(np.float64))
    assert m._stats['driver'] == 'gemhip_n2v_train_multi' and m._stats['n_gpus'] == 3 and m._stats['pairs'] > 0


def test_plugin_n_gpus_never_falls_back_to_one_gpu():
    """Two REAL ranks on the one-GPU box: the library's error comes through as GemHipError -- not a silent single-GPU run."""
    from gem_amd.embedding.gf import GraphFactorization
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('needs a box with ONE GPU')
    g = sbm_graph(512, 4000, 2, seed=1)
    with pytest.raises(_hip.GemHipError):
        GraphFactorization(d=8, eta=0.05, regu=0.01, max_iter=2, seed=1, n_gpus=2).learn_embedding(graph=g)
