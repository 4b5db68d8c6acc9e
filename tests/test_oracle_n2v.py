"""CPU tests: the node2vec oracle (oracle/n2v_oracle.c, oracle/snap_stream.py).  The reference binary is time-seeded and racy as
GEM runs it; with time() pinned (oracle/shim/faketime.c) and one thread it is deterministic, and BOTH halves are pinned to it at the
vector level: oracle/snap_stream.py (TRnd stream, Shuffle, PreprocessNode / GetNodeAlias in fp64, SimulateWalk; LearnEmbeddings /
TrainModel) reproduces walk matrices dumped from the running binary bit for bit and the embedding files the same runs wrote to the six
digits the binary prints (tests/golden/n2v_snap_stream_walks.json, scripts/make_golden_n2v_snap_stream.py).
The counter-based oracle the kernels are compared with (n2v_oracle.c: Philox instead of a sequential stream, rejection instead of
per-pair tables, fp32) is tied to that restatement: the same walk body and the same TrainModel body fed with its draws give its walks
bit for bit and its embeddings to fp32 rounding; alias targets equal; second-order frequencies follow the pinned tables.  Larger
graphs and Hogwild launches: MAP against the real binary (tests/golden/n2v_ref.json, produced by scripts/make_golden.py)."""
import ctypes as C
import json

import numpy as np
import pytest

import oracle
from gem_amd.embedding.node2vec import node2vec
from gem_amd.evaluation import reconstruction as gr
from gem_amd.graph import edge_arrays
from conftest import golden_path

SNAP = 11      # pad-zero | unigram quirk | uniform first hop


def chi2_ok(obs, exp_p, z=5.0):
    """Pearson chi-square against probabilities exp_p; accept within z sigma of its mean (dof)."""
    obs = np.asarray(obs, float); exp = exp_p * obs.sum()
    m = exp > 0
    assert np.all(obs[~m] == 0)
    chi2 = (((obs - exp) ** 2)[m] / exp[m]).sum()
    dof = m.sum() - 1
    return chi2 <= dof + z * np.sqrt(2 * max(dof, 1))


@pytest.mark.parametrize('n', [2, 3, 34, 1000, 1024, 1025, 65537])
def test_start_permutation_is_a_bijection(n):
    L = oracle.lib()
    for key in (1, 0xDEADBEEFCAFE):
        img = np.array([L.oracle_perm(j, n, key) for j in range(n)])
        assert np.array_equal(np.sort(img), np.arange(n))
    if n > 100:
        a = np.array([L.oracle_perm(j, n, 1) for j in range(n)]); b = np.array([L.oracle_perm(j, n, 2) for j in range(n)])
        assert (a == b).mean() < 0.05 and abs(np.corrcoef(a, np.arange(n))[0, 1]) < 0.1


def alias_implied(U, K):
    N = len(U)
    p = U.astype(np.float64).copy()
    for j in range(N):
        p[K[j]] += 1.0 - float(U[j])
    return p / N


def test_alias_tables_encode_the_weights():
    rng = np.random.RandomState(0)
    for N in (1, 2, 5, 64, 301):
        w = (rng.rand(N) ** 3 + 1e-3).astype(np.float32)
        U = np.zeros(N, np.float32); K = np.zeros(N, np.int32); work = np.zeros(N, np.int32)
        oracle.lib().oracle_alias_build_f32(N, oracle._p(w, C.c_float), oracle._p(U, C.c_float), oracle._p(K, C.c_int32),
                                            oracle._p(work, C.c_int32))
        assert np.all((U >= 0) & (U <= 1 + 1e-6))
        np.testing.assert_allclose(alias_implied(U, K), w / w.sum(), atol=2e-6)


def small_graph(weighted, seed=0, n=12, m=60):
    rng = np.random.RandomState(seed)
    key = np.unique(rng.randint(0, n, m) * n + rng.randint(0, n, m))
    src, dst = key // n, key % n
    keep = src != dst
    src, dst = src[keep].astype(np.int32), dst[keep].astype(np.int32)
    w = (rng.rand(len(src)) * 3 + 0.2).astype(np.float32) if weighted else None
    return n, src, dst, w


@pytest.mark.parametrize('p,q,weighted', [(1.0, 1.0, False), (1.0, 1.0, True), (0.25, 4.0, True), (4.0, 0.5, False)])
def test_walk_transitions_follow_node2vec_probabilities(p, q, weighted):
    """Empirical (prev, cur) -> next frequencies vs w(cur,x) * {1/p if x==prev; 1 if prev->x; 1/q else}
    (PreprocessNode), first hop uniform (SimulateWalk)."""
    n, src, dst, w = small_graph(weighted)
    row_ptr, col, ww = oracle.sorted_csr(n, src, dst, w)
    U = K = None
    if weighted:
        U, K = oracle.n2v_alias_rows(row_ptr, ww)
    walks = oracle.n2v_walks(row_ptr, col, U, K, p, q, 4000, 12, 77, SNAP)
    adj = set(zip(src.tolist(), dst.tolist()))
    wt = {(int(s), int(d)): (float(x) if w is not None else 1.0) for s, d, x in zip(src, dst, w if w is not None else np.ones(len(src)))}
    first = {}; second = {}
    for wk in walks:
        assert wk[1] == 0 or (wk[0], wk[1]) in adj or row_ptr[wk[0] + 1] == row_ptr[wk[0]]
        if row_ptr[wk[0] + 1] > row_ptr[wk[0]]:
            first.setdefault(int(wk[0]), []).append(int(wk[1]))
        for k in range(2, 12):
            t, v, x = int(wk[k - 2]), int(wk[k - 1]), int(wk[k])
            if row_ptr[v + 1] == row_ptr[v]:
                break
            assert (v, x) in adj
            second.setdefault((t, v), []).append(x)
    for v, xs in first.items():                          # first hop: uniform over out-neighbours
        nb = col[row_ptr[v]:row_ptr[v + 1]]
        assert chi2_ok([xs.count(int(x)) for x in nb], np.full(len(nb), 1.0 / len(nb)))
    checked = 0
    for (t, v), xs in second.items():
        if len(xs) < 400:
            continue
        nb = col[row_ptr[v]:row_ptr[v + 1]]
        a = np.array([wt[(v, int(x))] * (1 / p if x == t else (1.0 if (t, int(x)) in adj else 1 / q)) for x in nb])
        assert chi2_ok([xs.count(int(x)) for x in nb], a / a.sum()), (t, v)
        checked += 1
    assert checked >= 10


def test_walk_structure_and_padding(karate):
    n, src, dst, w, _ = edge_arrays(karate)
    row_ptr, col, ww = oracle.sorted_csr(n, src, dst, w)
    for flags, pad in ((SNAP, 0), (8, -1)):
        walks = oracle.n2v_walks(row_ptr, col, None, None, 1.0, 1.0, 10, 80, 5, flags)
        assert walks.shape == (340, 80)
        for r in range(10):                                  # every node starts exactly one walk per round
            assert np.array_equal(np.sort(walks[r * n:(r + 1) * n, 0]), np.arange(n))
        deg = np.diff(row_ptr)
        for wk in walks[:60]:
            k = 1
            while k < 80 and deg[wk[k - 1]] > 0:
                assert wk[k] in col[row_ptr[wk[k - 1]]:row_ptr[wk[k - 1] + 1]]
                k += 1
            assert np.all(wk[k:] == pad)                     # karate is a DAG (all edges i<j): walks die at sinks
        counts = oracle.n2v_vocab(n, walks)
        assert counts.sum() == (walks >= 0).sum()
    part = oracle.n2v_walks(row_ptr, col, None, None, 1.0, 1.0, 10, 80, 5, SNAP, 100, 230)     # shard == slice of the whole
    full = oracle.n2v_walks(row_ptr, col, None, None, 1.0, 1.0, 10, 80, 5, SNAP)
    assert np.array_equal(part, full[100:230])


def test_unigram_table_is_count_pow_075():
    counts = np.array([5, 0, 100, 37, 37, 1, 900, 12], np.int32)
    UT, KT = oracle.unigram_build(counts)
    want = counts.astype(float) ** 0.75
    np.testing.assert_allclose(alias_implied(UT, KT), want / want.sum(), atol=1e-6)


def test_oracle_map_matches_single_thread_snap(karate, sbm1024):
    """End to end: the sequential restatement lands where the real binary lands when it runs
    race-free (OMP_NUM_THREADS=1); the 8-thread binary is worse because its threads race on one RNG."""
    ref = json.load(open(golden_path('n2v_ref.json')))
    n, src, dst, w, _ = edge_arrays(sbm1024)
    maps = []
    for seed in (1, 2):
        X, _ = oracle.n2v_train(n, src, dst, w, 16, 80, 10, 10, 1, 1.0, 1.0, seed, SNAP)
        m = node2vec(d=16, max_iter=1, walk_len=80, num_walks=10, con_size=10, ret_p=1, inout_p=1)
        maps.append(gr.evaluateStaticGraphReconstruction(sbm1024, m, X.astype(np.float64), None)[0])
    t1 = np.mean(ref['sbm1024_d16_t1'])
    assert abs(np.mean(maps) - t1) <= 0.03 * t1, (maps, ref['sbm1024_d16_t1'])
    assert np.mean(maps) > np.mean(ref['sbm1024_d16_t8'])
    n, src, dst, w, _ = edge_arrays(karate)
    maps = []
    for seed in range(8):
        X, _ = oracle.n2v_train(n, src, dst, w, 2, 80, 10, 10, 1, 1.0, 1.0, seed, SNAP)
        m = node2vec(d=2, max_iter=1, walk_len=80, num_walks=10, con_size=10, ret_p=1, inout_p=1)
        maps.append(gr.evaluateStaticGraphReconstruction(karate, m, X.astype(np.float64), None)[0])
    lo = min(ref['karate_d2_t1'] + ref['karate_d2_t8']) - 0.1; hi = max(ref['karate_d2_t1'] + ref['karate_d2_t8']) + 0.1
    assert lo <= np.mean(maps) <= hi, maps
    # and the reference's own acceptance test (tests/test_karate.py:57-60,78) holds for the restatement
    tgt = np.loadtxt(golden_path('ref_karate_node2vec.txt'))
    assert abs(np.mean(tgt - X)) < 0.3


# ------------------------------------------------------------------ vector-level pin of the walk half to the reference binary
def _stream_cases():
    return json.load(open(golden_path('n2v_snap_stream_walks.json')))['cases']


@pytest.mark.parametrize('name', ['karate_p1_q1', 'karate_p0.25_q4', 'karate_p4_q0.25', 'karate_weighted_p0.5_q2',
                                  'directed_with_sinks_p1_q1', 'directed_with_sinks_p2_q0.5', 'karate_two_epochs_alpha_schedule'])
def test_snap_stream_restatement_reproduces_the_reference_binary_walk_for_walk(name):
    """gem/c_exe/node2vec with time() pinned and OMP_NUM_THREADS=1, its walk matrix dumped at LearnEmbeddings() entry, against
    oracle/snap_stream.py on the same edge file, (p, q) and seed: every token of every walk, including the zero padding behind sinks."""
    from oracle import snap_stream as ss
    c = _stream_cases()[name]
    order, nbr, w = ss.load_edge_list(c['edge_lines'], directed=True, weighted=True)
    mine = ss.simulate_walks(order, nbr, w, c['p'], c['q'], c['num_walks'], c['walk_len'], c['seed'])
    ref = np.asarray(c['walks'], dtype=np.int32)
    assert mine.shape == ref.shape and np.array_equal(mine, ref)
    if 'sinks' in name:
        assert (ref[:, -1] == 0).sum() > 5                       # the case does exercise early stops


def test_counter_based_oracle_builds_the_alias_tables_the_pinned_restatement_builds():
    """GetNodeAlias: oracle_alias_build_f32 (fp32, what the device builds bit for bit) against snap_stream.node_alias (fp64, pinned to the
    binary through its walks): the same alias targets K -- same stacks, filled in index order, popped from the back -- and U to fp32."""
    from oracle import snap_stream as ss
    rng = np.random.RandomState(3)
    for N in (1, 2, 3, 7, 33, 200):
        w = (rng.rand(N) ** 2 + 0.05).astype(np.float32)
        U = np.zeros(N, np.float32); K = np.zeros(N, np.int32); work = np.zeros(N, np.int32)
        oracle.lib().oracle_alias_build_f32(N, oracle._p(w, C.c_float), oracle._p(U, C.c_float), oracle._p(K, C.c_int32), oracle._p(work, C.c_int32))
        P = w.astype(np.float64) / w.astype(np.float64).sum()
        K64, U64 = ss.node_alias(P.tolist())
        if np.abs(np.asarray(U64) - 1.0).min() > 1e-5 or N == 1:      # (a U within rounding of 1 may land on the other stack in fp32)
            assert K.tolist() == K64
            np.testing.assert_allclose(U, U64, atol=3e-6)
        np.testing.assert_allclose(alias_implied(U, K), alias_implied(np.asarray(U64), K64), atol=2e-6)


def test_hub_rows_closed_form_vose_is_get_node_alias():
    """Rows of >= 2048 neighbours take the closed form of GetNodeAlias's loop (two prefix sums in stack order + binary searches: what the device's
    n2v_alias_hub_kernel runs with a workgroup per row, same fixed summation order, bit for bit).  Against snap_stream.node_alias -- the sequential
    loop in fp64, pinned to the binary through its walks: the SAME alias target for every entry that has one, U to fp32 rounding, the same entries at
    U = 1, and the table encodes the weights.  Shapes: uniform, heavy-tailed (a hub's neighbours on a weighted power-law graph), all equal, one giant
    weight among dust, and a size that is not a multiple of the 256-entry chunk."""
    from oracle import snap_stream as ss
    rng = np.random.RandomState(11)
    cases = [(2048, rng.rand(2048)), (5000, rng.pareto(1.1, 5000) + 0.01), (20011, rng.pareto(0.9, 20011) + 1e-3), (3000, np.ones(3000)),
             (4097, np.where(np.arange(4097) == 7, 100.0, 1e-3))]
    for N, w in cases:
        w = w.astype(np.float32)
        U = np.zeros(N, np.float32); K = np.zeros(N, np.int32)
        oracle.lib().oracle_alias_build_hub(N, oracle._p(w, C.c_float), oracle._p(U, C.c_float), oracle._p(K, C.c_int32))
        P = w.astype(np.float64) / w.astype(np.float64).sum()
        K64, U64 = ss.node_alias(P.tolist())
        K64 = np.asarray(K64); U64 = np.asarray(U64)
        has = U64 < 1.0
        assert np.array_equal(K[has], K64[has])
        np.testing.assert_allclose(U, U64, atol=1e-7)
        assert np.array_equal(U64 >= 1.0, U.astype(np.float64) >= 1.0 - 1e-7)
        np.testing.assert_allclose(alias_implied(U, K), P, atol=2e-7)
    # oracle_n2v_alias_rows switches per row: a CSR with one hub row and short rows
    deg = [5, 3000, 0, 17]
    row_ptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    ww = (rng.pareto(1.3, row_ptr[-1]) + 0.05).astype(np.float32)
    U, K = oracle.n2v_alias_rows(row_ptr, ww)
    for v, dg in enumerate(deg):
        if dg == 0:
            continue
        a = row_ptr[v]
        np.testing.assert_allclose(alias_implied(U[a:a + dg], K[a:a + dg]), ww[a:a + dg].astype(np.float64) / ww[a:a + dg].astype(np.float64).sum(), atol=3e-6)


def test_counter_based_oracle_walks_follow_the_pinned_transition_tables():
    """The rejection sampler of n2v_oracle.c (what the HIP walk kernel equals bit for bit) against the per-(t, v) alias tables of the
    restatement that reproduces the binary: the probabilities those tables encode are what its (t, v) -> x frequencies must follow."""
    from oracle import snap_stream as ss
    c = _stream_cases()['karate_weighted_p0.5_q2']
    order, nbr, w = ss.load_edge_list(c['edge_lines'], directed=True, weighted=True)
    tables = ss.preprocess_transition_probs(order, nbr, w, c['p'], c['q'])
    e = np.array([[int(f) for f in ln.split()[:2]] for ln in c['edge_lines']])
    wt = np.array([float(ln.split()[2]) for ln in c['edge_lines']], dtype=np.float32)
    n = int(e.max()) + 1
    row_ptr, col, ww = oracle.sorted_csr(n, e[:, 0], e[:, 1], wt)
    U, K = oracle.n2v_alias_rows(row_ptr, ww)
    walks = oracle.n2v_walks(row_ptr, col, U, K, c['p'], c['q'], 1500, 20, 5, SNAP)
    second = {}
    for wk in walks:
        for k in range(2, walks.shape[1]):
            second.setdefault((int(wk[k - 2]), int(wk[k - 1])), []).append(int(wk[k]))
    checked = 0
    for (t, v), xs in second.items():
        if len(xs) < 1500:
            continue
        Kt, Ut = tables[(t, v)]
        probs = alias_implied(np.asarray(Ut), Kt)
        assert chi2_ok([xs.count(x) for x in nbr[v]], probs), (t, v)
        checked += 1
    assert checked >= 20


@pytest.mark.parametrize('name', ['karate_p1_q1', 'karate_p0.25_q4', 'karate_weighted_p0.5_q2', 'directed_with_sinks_p2_q0.5',
                                  'karate_two_epochs_alpha_schedule'])
def test_snap_stream_restatement_reproduces_the_embedding_file_the_reference_binary_wrote(name):
    """The SGNS half: from the binary's own walks, oracle/snap_stream.learn_embeddings (renaming, second TRnd, InitPosEmb, unigram alias,
    RndUnigramInt's quirk, TrainModel with the exp table) against the .emb file of the SAME deterministic run -- same node order, every
    number to the six significant digits `%g` prints (relative 5e-6).  The last case runs two epochs over 10 200 words each: the alpha
    schedule (refreshed every 10 000 words against epochs * words + 1) is part of what it pins."""
    from oracle import snap_stream as ss
    c = _stream_cases()[name]
    ids, X = ss.learn_embeddings(c['walks'], c['d'], c['window'], c['epochs'], c['seed'])
    lines = c['emb'].strip().split('\n')
    assert lines[0].split() == [str(len(ids)), str(c['d'])]
    rows = [ln.split() for ln in lines[1:]]
    assert [int(r[0]) for r in rows] == ids
    R = np.array([[float(x) for x in r[1:]] for r in rows])
    assert np.all(np.abs(X - R) <= 5.5e-6 * np.abs(R) + 1e-12), float((np.abs(X - R) / np.abs(R)).max())
    assert np.abs(R).max() > 0.1                                  # (the run did train: rows moved far from their +-1/16 initial range)


def pinned_sgns_on_kernel_draws(walks, n, d, window, seed, UT, KT):
    """snap_stream.train_model -- the TrainModel body that reproduces the binary -- in fp64 with the exact sigmoid, fed with the COUNTER-BASED
    draws of n2v_oracle.c / the HIP kernels (Philox: window shrink from (walk, pos), negatives from (walk, pos, slot, sample), SNAP's
    RndUnigramInt quirk on the fp32 unigram table given) from the initial tables every implementation draws for `seed`.  Returns (SynPos, SynNeg)."""
    from oracle import snap_stream as ss
    P0, N0 = oracle.sgns_init(n, d, seed)
    P64, N64 = P0.astype(np.float64), N0.astype(np.float64)
    L = oracle.lib()
    buf = (C.c_uint32 * 4)()

    def philox(c0, c1, c2, c3):
        L.oracle_philox(C.c_uint64(seed), C.c_uint32(c0), C.c_uint32(c1), C.c_uint32(c2), C.c_uint32(c3), buf)
        return buf[0], buf[1]

    def offset_draw(wi, pos):
        return philox(wi, 0, pos, 2)[0] % window                      # TAG_WIN, epoch 0

    def negative_draw(wi, pos, a, j):
        x, y = philox(wi, 0, pos | (a << 16), 3 | (j << 16))          # TAG_NEG, epoch 0
        X = int(KT[(x * n) >> 32])                                    # SNAP's quirk: the alias of the slot
        return X if np.float32(y >> 8) * np.float32(1.0 / 16777216.0) < UT[X] else int(KT[X])
    ss.train_model(np.asarray(walks, dtype=np.int64), P64, N64, window, 1, offset_draw, negative_draw, sigmoid='exact')
    return P64, N64


def test_counter_based_sgns_oracle_is_the_pinned_train_model_on_other_draws():
    """oracle_sgns_train (fp32, Philox draws; what the deterministic HIP launch is compared with) against snap_stream.train_model -- the
    body that reproduces the binary's output -- fed with the SAME Philox draws: same window shrinks, same negative targets (incl.
    RndUnigramInt's quirk), same update order; what remains is fp32 against fp64.  (tests/test_n2v_gpu.py runs the same comparison
    with the HIP kernels in the oracle's place.)"""
    c = _stream_cases()['karate_p1_q1']
    e = np.array([[int(f) for f in ln.split()[:2]] for ln in c['edge_lines']])
    n, d, window, seed = int(e.max()) + 1, 8, 3, 12345
    row_ptr, col, _ = oracle.sorted_csr(n, e[:, 0], e[:, 1], None)
    walks = oracle.n2v_walks(row_ptr, col, None, None, 1.0, 1.0, 3, 12, seed, SNAP)
    UT, KT = oracle.unigram_build(oracle.n2v_vocab(n, walks))
    P, N = oracle.sgns_init(n, d, seed)
    oracle.sgns_train(walks, window, 0.025, 1, 0, walks.size, 0, 0, UT, KT, seed, SNAP, P, N)
    P64, N64 = pinned_sgns_on_kernel_draws(walks, n, d, window, seed, UT, KT)
    assert np.abs(P - P64).max() <= 2e-5 * np.abs(P64).max() and np.abs(N - N64).max() <= 2e-5 * np.abs(N64).max()
    assert np.abs(P64).max() > 0.1


def pinned_walks_on_kernel_draws(edge_lines, seed, rounds, walk_len):
    """snap_stream.simulate_walks -- the walk body that reproduces the binary -- fed with the COUNTER-BASED draws of n2v_oracle.c / the HIP walk
    kernel (Feistel start permutation per round, Philox (walk, length) for the neighbour slot), first-order unweighted case.  Returns the matrix."""
    from oracle import snap_stream as ss
    order, nbr, w = ss.load_edge_list(edge_lines, directed=True, weighted=True)
    e = np.array([[int(f) for f in ln.split()[:2]] for ln in edge_lines])
    n = int(e.max()) + 1
    row_ptr, col, _ = oracle.sorted_csr(n, e[:, 0], e[:, 1], None)
    lib = oracle.lib()
    start = oracle.start_nodes(row_ptr, col)
    m = len(start)
    assert m == len(order)
    buf = (C.c_uint32 * 4)()

    class Draws(object):
        def round_order(self, r, ids):
            key = seed ^ (((r + 1) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
            return [int(start[lib.oracle_perm(j, m, C.c_uint64(key))]) for j in range(m)]

        def _slot(self, wid, length, deg):
            lib.oracle_philox(C.c_uint64(seed), C.c_uint32(wid), C.c_uint32(0), C.c_uint32(length), C.c_uint32(1), buf)   # TAG_WALK, trial 0
            return (buf[0] * deg) >> 32

        def first_hop(self, wid, deg):
            return self._slot(wid, 1, deg)

        def alias_draw(self, wid, length, tab, dst):
            assert all(abs(u - 1.0) < 1e-12 for u in tab[1])              # unweighted, p = q = 1: every slot accepts
            return self._slot(wid, length, len(tab[0]))
    return ss.simulate_walks(order, nbr, w, 1.0, 1.0, rounds, walk_len, seed, draws=Draws())


@pytest.mark.parametrize('name', ['karate_p1_q1', 'directed_with_sinks_p1_q1'])
def test_counter_based_walk_oracle_is_the_pinned_simulate_walk_on_other_draws(name):
    """oracle_n2v_walks (what the HIP walk kernel equals bit for bit) against snap_stream.simulate_walks -- the body that reproduces the
    binary's walks -- fed with the SAME Philox draws and the same Feistel start permutation, first-order unweighted case (the headline's):
    identical matrices, sinks and zero padding included.  (tests/test_n2v_gpu.py runs the same comparison with the HIP kernel itself.)"""
    c = _stream_cases()[name]
    e = np.array([[int(f) for f in ln.split()[:2]] for ln in c['edge_lines']])
    n, seed, rounds, L_ = int(e.max()) + 1, 424242, 3, 14
    row_ptr, col, _ = oracle.sorted_csr(n, e[:, 0], e[:, 1], None)
    ref = oracle.n2v_walks(row_ptr, col, None, None, 1.0, 1.0, rounds, L_, seed, SNAP)
    assert np.array_equal(pinned_walks_on_kernel_draws(c['edge_lines'], seed, rounds, L_), ref)


def test_counter_based_weighted_walks_are_the_pinned_body_on_other_draws():
    """The same tie on a WEIGHTED graph at p = q = 1: uniform first hop (SimulateWalk ignores the weights there), then slot / accept / alias
    against the first-order table of the current node -- n2v_oracle.c's fp32 per-row tables stand in for the binary's per-pair tables, which
    at p = q = 1 encode the same row distribution (equal alias targets for generic weights: test above; with small-integer weights some U
    land exactly on 1 and the two precisions may stack them differently -- another valid table of the same distribution)."""
    from oracle import snap_stream as ss
    c = _stream_cases()['karate_weighted_p0.5_q2']
    order, nbr, w = ss.load_edge_list(c['edge_lines'], directed=True, weighted=True)
    e = np.array([[int(f) for f in ln.split()[:2]] for ln in c['edge_lines']])
    wt = np.array([float(ln.split()[2]) for ln in c['edge_lines']], dtype=np.float32)
    n, seed, rounds, L_ = int(e.max()) + 1, 777, 4, 16
    row_ptr, col, ww = oracle.sorted_csr(n, e[:, 0], e[:, 1], wt)
    U, K = oracle.n2v_alias_rows(row_ptr, ww)
    ref = oracle.n2v_walks(row_ptr, col, U, K, 1.0, 1.0, rounds, L_, seed, SNAP)
    lib = oracle.lib()
    start = oracle.start_nodes(row_ptr, col)
    m = len(start)
    buf = (C.c_uint32 * 4)()

    class Draws(object):
        def round_order(self, r, ids):
            key = seed ^ (((r + 1) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
            return [int(start[lib.oracle_perm(j, m, C.c_uint64(key))]) for j in range(m)]

        def _draw(self, wid, length):
            lib.oracle_philox(C.c_uint64(seed), C.c_uint32(wid), C.c_uint32(0), C.c_uint32(length), C.c_uint32(1), buf)
            return buf[0], buf[1]

        def first_hop(self, wid, deg):
            return (self._draw(wid, 1)[0] * deg) >> 32

        def alias_draw(self, wid, length, tab, dst):
            a, deg = int(row_ptr[dst]), len(tab[0])
            x, y = self._draw(wid, length)
            slot = (x * deg) >> 32
            return slot if np.float32(y >> 8) * np.float32(1.0 / 16777216.0) < U[a + slot] else int(K[a + slot])
    mine = ss.simulate_walks(order, nbr, w, 1.0, 1.0, rounds, L_, seed, draws=Draws())
    assert np.array_equal(mine, ref)


def test_c_port_of_the_pinned_restatement_equals_it():
    """oracle/snap_stream.c against oracle/snap_stream.py on every small case: walks token for token, embeddings to rounding of the exp
    table (numpy's exp against libm's pow)."""
    from oracle import snap_stream as ss
    for name, c in _stream_cases().items():
        order, nbr, w = ss.load_edge_list(c['edge_lines'], directed=True, weighted=True)
        assert np.array_equal(ss.fast_walks(order, nbr, w, c['p'], c['q'], c['num_walks'], c['walk_len'], c['seed']), np.asarray(c['walks'], dtype=np.int32)), name
        if c['epochs'] == 1:
            ids, X = ss.fast_learn_embeddings(c['walks'], c['d'], c['window'], c['epochs'], c['seed'])
            ids2, X2 = ss.learn_embeddings(c['walks'], c['d'], c['window'], c['epochs'], c['seed'])
            assert ids == ids2 and np.abs(X - X2).max() <= 1e-9, name


def test_pinned_restatement_reproduces_the_binary_at_the_reference_hyper_parameters():
    """The reference's own SBM-1024 graph, walk_len 80, num_walks 10, con_size 10, p = q = 1 (examples/run_sbm.py:70) at d = 128:
    819 200 words, 82 refreshes of alpha, both clamp branches of the sigmoid.  gem/c_exe/node2vec under the time() shim against
    oracle/snap_stream.c: the 10 240 x 80 walk matrix bit for bit (SHA-256), the node order of the embedding file, and each of its
    131 072 numbers to the six digits the binary prints."""
    import hashlib
    from oracle import snap_stream as ss
    g = np.load(golden_path('n2v_snap_stream_sbm1024.npz'))
    pr = json.loads(str(g['params']))
    e = np.load(golden_path('sbm1024_edges.npy'))
    order, nbr, w = ss.load_edge_list(['%d %d %f' % (int(i), int(j), 1.0) for i, j in e.tolist()], directed=True, weighted=True)
    walks = ss.fast_walks(order, nbr, w, pr['p'], pr['q'], pr['num_walks'], pr['walk_len'], pr['seed'])
    assert walks.shape == tuple(g['walks_shape']) and np.array_equal(walks[:4], g['walks_head'])
    assert hashlib.sha256(walks.tobytes()).hexdigest() == str(g['walks_sha256'])
    ids, X = ss.fast_learn_embeddings(walks, pr['d'], pr['window'], pr['epochs'], pr['seed'])
    R = g['emb'].astype(np.float64)                                        # (stored as float32: 6e-8 on top of the file's 5e-6)
    assert np.array_equal(ids, g['ids'])
    assert np.all(np.abs(X - R) <= 5.7e-6 * np.abs(R) + 1e-12), float((np.abs(X - R) / np.abs(R)).max())
    assert np.abs(R).max() > 0.5
