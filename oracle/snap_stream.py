"""oracle/snap_stream.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Sequential restatement of BOTH halves (walks, then skip-gram with negative sampling) of the reference's node2vec path -- the prebuilt SNAP binary gem/c_exe/node2vec that
gem/embedding/node2vec.py:34-48 runs -- INCLUDING ITS RANDOM STREAM, so that it can be compared with the binary walk for walk.
The binary's source is third-party and absent from /root/reference (snap-stanford/snap, examples/node2vec + snap-adv/n2v.cpp,
biasedrandomwalk.cpp; ELF banner "Apr 9 2017", not stripped); what is restated here was read off the ELF's symbols and
disassembly and then PINNED against the binary itself, made deterministic with oracle/shim/faketime.c (time() fixed) and
OMP_NUM_THREADS=1: tests/golden/n2v_snap_stream_walks.json holds walk matrices dumped from the running binary
(scripts/make_golden_n2v_snap_stream.py), tests/test_oracle_n2v.py asserts this module reproduces them bit for bit -- and, from
those walks, the embedding file the same run wrote, to the six significant digits the binary prints (%g).

    TRnd::GetUniDevInt  @0x41baf0   Seed <- 16807*(Seed % 127773) - 2836*(Seed / 127773), +2147483647 if <= 0; value % Range
    TRnd::GetUniDev                 next Seed / 2147483647.0
    TRnd::PutSeed       @0x41b9a0   a non-zero seed is stored as it is (the binary passes time(NULL))
    TVec::Shuffle                   for i in 0..n-2: swap(i, i + GetUniDevInt(n - i))
    graph loader                    nodes are added in order of first appearance in the edge file (source, then destination);
                                    out-neighbours are kept sorted by id; `-dr` = directed, `-w` = third column is the weight
    node2vec()          @0x40c420   TRnd Rnd(time(NULL)); per round: NIdsV.Shuffle(Rnd), then SimulateWalk for every node in
                                    that order; walks fill row round*N + j of a zero-initialised matrix
    SimulateWalk        @0x411a00   first hop UNIFORM: GetNbrNId(GetUniDevInt(deg)); then AliasDrawInt on the table of the pair
                                    (previous, current); stops at a node without out-neighbours (row stays zero-padded)
    AliasDrawInt        @0x411360   X = int(GetUniDev() * N); Y = GetUniDev(); Y < U[X] ? X : K[X]      (always two draws)
    PreprocessNode      @0x411f40   for t -> v -> x:  w(v,x)/p if x == t,  w(v,x) if x is an out-neighbour of t,  else w(v,x)/q;
                                    normalised by their sum, then GetNodeAlias
    GetNodeAlias        @0x4115f0   Vose: U[i] = P[i]*N, under/over stacks filled in index order and popped from the BACK,
                                    K initialised to 0, leftovers get U = 1                              (all in fp64)
    LearnEmbeddings     @0x40ea30   tokens renamed to 0..N-1 by first appearance; a SECOND TRnd(time(NULL)); InitPosEmb @0x40e270
                                    (u - 0.5)/d row by row; InitNegEmb @0x40e040 zeros; InitUnigramTable @0x40e520 count^0.75, Vose
    RndUnigramInt       @0x40d5f0   X = KTable[int(u*n)] (sic), Y = u'; Y < UTable[X] ? X : KTable[X]
    TrainModel          @0x40d6a0   alpha refreshed when WordCntAll % 10000 == 0; b = GetUniDevInt() % window; context slots
                                    b .. 2*window-b; 1 + 5 targets (a negative equal to the word is skipped); exp table of 120 000
                                    entries over [-6, 6), index truncated; SynNeg(target) and the context's SynPos row updated
                                    as in word2vec.c; embeddings written in renamed order with %g                 (all in fp64)

How this ties the product to the reference: oracle/n2v_oracle.c and the HIP kernels implement the SAME walk semantics (uniform
first hop, alias pick among sorted neighbours, zero padding, start nodes = nodes that occur in the edge list, a fresh
permutation of them per round) on a counter-based stream (Philox) instead of TRnd's sequential one -- a GPU cannot consume a
sequential stream -- and, for p,q != 1, by rejection from the first-order table instead of the per-pair tables, with the same
target probabilities (the weights of PreprocessNode above; chi-square test in tests/test_n2v_gpu.py).  Pure-Python loops: small
graphs only.
"""
import numpy as np

M, A, Q, R = 2147483647, 16807, 127773, 2836


class TRnd(object):
    def __init__(self, seed):
        assert seed > 0
        self.seed = int(seed)

    def _next(self):
        v = A * (self.seed % Q) - R * (self.seed // Q)
        if v <= 0:
            v += M
        self.seed = v
        return v

    def uni_dev(self):
        return self._next() / float(M)

    def uni_dev_int(self, rng=0):
        v = self._next()
        return v if rng == 0 else v % rng


def shuffle(v, rnd):
    n = len(v)
    for i in range(n - 1):
        j = i + rnd.uni_dev_int(n - i)
        v[i], v[j] = v[j], v[i]


def load_edge_list(lines, directed=True, weighted=True):
    """(node order of first appearance, {v: sorted out-neighbour ids}, {(v, x): weight}) the way the binary's loader builds its net."""
    order, seen, w = [], set(), {}
    for ln in lines:
        f = ln.split()
        if len(f) < 2:
            continue
        i, j = int(f[0]), int(f[1])
        wt = float(f[2]) if (weighted and len(f) > 2) else 1.0
        for x in (i, j):
            if x not in seen:
                seen.add(x)
                order.append(x)
        w[(i, j)] = wt
        if not directed:
            w[(j, i)] = wt
    nbr = {v: [] for v in order}
    for (i, j) in w:
        nbr[i].append(j)
    for v in nbr:
        nbr[v].sort()
    return order, nbr, w


def node_alias(P):
    """GetNodeAlias: P normalised probabilities (fp64) -> (K, U)."""
    N = len(P)
    K = [0] * N
    U = [0.0] * N
    under, over = [], []
    for i in range(N):
        U[i] = P[i] * N
        (under if U[i] < 1 else over).append(i)
    while under and over:
        s, l = under.pop(), over.pop()
        K[s] = l
        U[l] = U[l] + U[s] - 1
        (under if U[l] < 1 else over).append(l)
    while under:
        U[under.pop()] = 1.0
    while over:
        U[over.pop()] = 1.0
    return K, U


def preprocess_transition_probs(order, nbr, w, p, q):
    """{(t, v): (K, U)} for every edge t -> v (PreprocessNode)."""
    tables = {}
    for t in order:
        nt = set(nbr[t])
        for v in nbr[t]:
            P, s = [], 0.0
            for x in nbr[v]:
                wt = w[(v, x)]
                a = wt / p if x == t else (wt if x in nt else wt / q)
                P.append(a)
                s += a
            tables[(t, v)] = node_alias([a / s for a in P])
    return tables


def alias_draw_int(tab, rnd):
    K, U = tab
    x = int(rnd.uni_dev() * len(K))
    y = rnd.uni_dev()
    return x if y < U[x] else K[x]


def simulate_walks(order, nbr, w, p, q, num_walks, walk_len, seed, draws=None):
    """The walk matrix [num_walks * N][walk_len] (int32, zero padded) node2vec() hands to LearnEmbeddings.
    draws=None: the binary's TRnd stream.  Otherwise an object supplying the three random decisions -- round_order(r, ids) -> start
    nodes of round r, first_hop(walk, deg) -> neighbour index, alias_draw(walk, length, (K, U), current node) -> neighbour index -- so that the SAME
    body runs on the counter-based draws of oracle/n2v_oracle.c (tests/test_oracle_n2v.py ties that file's walks to this one)."""
    tables = preprocess_transition_probs(order, nbr, w, float(p), float(q))
    rnd = TRnd(seed) if draws is None else None
    ids = list(order)
    n = len(ids)
    out = np.zeros((num_walks * n, walk_len), dtype=np.int32)
    for r in range(num_walks):
        if draws is None:
            shuffle(ids, rnd)
        else:
            ids = draws.round_order(r, ids)
        for j, s in enumerate(ids):
            wid = r * n + j
            wk = [s]
            if walk_len > 1 and nbr[s]:
                wk.append(nbr[s][rnd.uni_dev_int(len(nbr[s])) if draws is None else draws.first_hop(wid, len(nbr[s]))])
                while len(wk) < walk_len:
                    dst, src = wk[-1], wk[-2]
                    if not nbr[dst]:
                        break
                    tab = tables[(src, dst)]
                    wk.append(nbr[dst][alias_draw_int(tab, rnd) if draws is None else draws.alias_draw(wid, len(wk), tab, dst)])
            out[wid, :len(wk)] = wk
    return out


# ------------------------------------------------------------------------------------------ LearnEmbeddings (the SGNS half)
MAX_EXP, EXP_TABLE_PRECISION, NEG_SAM_N, START_ALPHA = 6, 10000, 5, 0.025
TABLE_SIZE = MAX_EXP * EXP_TABLE_PRECISION * 2


def unigram_table(vocab):
    """InitUnigramTable @0x40e520: count^0.75 normalised, Vose in fp64 with the same stack discipline as GetNodeAlias."""
    p = [float(c) ** 0.75 for c in vocab]
    s = 0.0
    for x in p:
        s += x
    return node_alias([x / s for x in p])


def rnd_unigram_int(K, U, rnd):
    """RndUnigramInt @0x40d5f0: X = KTable[int(u * n)] (sic: the alias of a random slot), Y = u'; Y < UTable[X] ? X : KTable[X]."""
    x = K[int(rnd.uni_dev() * len(K))]
    y = rnd.uni_dev()
    return x if y < U[x] else K[x]


def train_model(walks, syn_pos, syn_neg, window, iters, offset_draw, negative_draw, sigmoid='table', alpha0=START_ALPHA):
    """TrainModel @0x40d6a0, walk by walk on one thread, in place on syn_pos / syn_neg (fp64 [N][d]).  The two random decisions are
    callables so that the SAME body runs on the binary's TRnd stream (learn_embeddings below: pinned to the binary's output) and on
    the counter-based draws of oracle/n2v_oracle.c (tests/test_oracle_n2v.py: ties that file's fp32 restatement to this one):
        offset_draw(walk, pos)            -> window shrink b in [0, window)
        negative_draw(walk, pos, a, j)    -> negative target for context slot a, sample j = 1..5
    sigmoid 'table' = the binary's 120 000-entry exp table (index truncated towards zero), 'exact' = 1 / (1 + e^x)."""
    d = syn_pos.shape[1]
    exp_table = np.exp(-MAX_EXP + np.arange(TABLE_SIZE) / float(EXP_TABLE_PRECISION))     # (the binary: pow(e, value))
    all_words = walks.shape[0] * walks.shape[1]
    alpha = alpha0
    cnt = 0
    L = walks.shape[1]
    for _ in range(iters):
        for wi in range(walks.shape[0]):
            wk = walks[wi]
            for pos in range(L):
                if cnt % 10000 == 0:
                    alpha = alpha0 * (1 - cnt / float(iters * all_words + 1))
                    if alpha < alpha0 * 0.0001:
                        alpha = alpha0 * 0.0001
                word = int(wk[pos])
                off = offset_draw(wi, pos)
                for a in range(off, window * 2 + 1 - off):
                    if a == window:
                        continue
                    cp = pos - window + a
                    if cp < 0 or cp >= L:
                        continue
                    cur = int(wk[cp])
                    neu1e = np.zeros(d)
                    for j in range(NEG_SAM_N + 1):
                        if j == 0:
                            target, label = word, 1
                        else:
                            target = negative_draw(wi, pos, a, j)
                            if target == word:
                                continue
                            label = 0
                        prod = 0.0
                        for k in range(d):
                            prod += syn_pos[cur, k] * syn_neg[target, k]
                        if prod > MAX_EXP:
                            g = (label - 1) * alpha
                        elif prod < -MAX_EXP:
                            g = label * alpha
                        elif sigmoid == 'table':
                            g = (label - 1 + 1 / (1 + exp_table[int(prod * EXP_TABLE_PRECISION) + TABLE_SIZE // 2])) * alpha
                        else:
                            g = (label - 1 + 1 / (1 + np.exp(prod))) * alpha
                        for k in range(d):
                            neu1e[k] += g * syn_neg[target, k]
                            syn_neg[target, k] += g * syn_pos[cur, k]
                    syn_pos[cur] += neu1e
                cnt += 1


def learn_embeddings(walks, d, window, iters, seed):
    """LearnEmbeddings @0x40ea30 on one thread: tokens renamed to 0..N-1 in order of first appearance (the zero padding behind a sink
    is renamed like node 0), LearnVocab, TRnd Rnd(time(NULL)), InitPosEmb ((u - 0.5) / d, row by row), InitNegEmb (zeros),
    InitUnigramTable, then TrainModel walk by walk; per word one GetUniDevInt() % window, per negative sample the two GetUniDev of
    RndUnigramInt.  Returns (node ids in the order the binary writes them, fp64 [N][d] = the rows it writes with %g)."""
    walks = np.array(walks, dtype=np.int64)
    rnm, back = {}, []
    for i in range(walks.shape[0]):
        for j in range(walks.shape[1]):
            v = int(walks[i, j])
            if v not in rnm:
                rnm[v] = len(back)
                back.append(v)
            walks[i, j] = rnm[v]
    n = len(back)
    vocab = np.bincount(walks.ravel(), minlength=n)
    rnd = TRnd(seed)
    syn_pos = np.zeros((n, d))
    for i in range(n):
        for j in range(d):
            syn_pos[i, j] = (rnd.uni_dev() - 0.5) / d
    syn_neg = np.zeros((n, d))
    K, U = unigram_table(vocab.tolist())
    train_model(walks, syn_pos, syn_neg, window, iters, lambda wi, pos: rnd.uni_dev_int() % window,
                lambda wi, pos, a, j: rnd_unigram_int(K, U, rnd))
    return back, syn_pos


# ------------------------------------------------------------------------------------------ C port (oracle/snap_stream.c)
def _clib():
    import ctypes as C
    import oracle
    L = oracle.lib()
    if not getattr(L, '_snap_stream_ready', False):
        i32p, i64p, f64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double)
        L.snap_stream_walks.restype = C.c_int
        L.snap_stream_walks.argtypes = [C.c_int64, C.c_int64, i32p, i64p, i32p, f64p, C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_int32, i32p]
        L.snap_stream_learn_embeddings.restype = C.c_int
        L.snap_stream_learn_embeddings.argtypes = [C.c_int64, C.c_int32, i32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, i32p, f64p, i64p]
        L._snap_stream_ready = True
    return L


def fast_walks(order, nbr, w, p, q, num_walks, walk_len, seed):
    """simulate_walks through the C port (same arguments, same result)."""
    import ctypes as C
    n = max(order) + 1
    deg = np.zeros(n, dtype=np.int64)
    for v in order:
        deg[v] = len(nbr[v])
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(deg, out=row_ptr[1:])
    col = np.zeros(max(int(row_ptr[-1]), 1), dtype=np.int32)
    wt = np.zeros(max(int(row_ptr[-1]), 1), dtype=np.float64)
    for v in order:
        a = int(row_ptr[v])
        for k, x in enumerate(nbr[v]):
            col[a + k] = x
            wt[a + k] = w[(v, x)]
    ordr = np.ascontiguousarray(order, dtype=np.int32)
    out = np.zeros((num_walks * len(order), walk_len), dtype=np.int32)
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    rc = _clib().snap_stream_walks(n, len(order), P(ordr, C.c_int32), P(row_ptr, C.c_int64), P(col, C.c_int32), P(wt, C.c_double), float(p), float(q),
                                   num_walks, walk_len, int(seed), P(out, C.c_int32))
    assert rc == 0
    return out


def fast_learn_embeddings(walks, d, window, iters, seed, rename=True):
    """learn_embeddings through the C port (same arguments, same result).  rename=False: an experiment, not the binary -- tokens keep
    their node ids, so the unigram alias table is laid out in node-id order (scripts/unigram_layout_effect.py)."""
    import ctypes as C
    wk = np.ascontiguousarray(walks, dtype=np.int32).copy()
    bound = int(wk.max()) + 1
    ids = np.zeros(bound, dtype=np.int32)
    emb = np.zeros((bound, d), dtype=np.float64)
    n_out = C.c_int64()
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    rc = _clib().snap_stream_learn_embeddings(wk.shape[0], wk.shape[1], P(wk, C.c_int32), bound, d, window, iters, int(seed), 1 if rename else 0, P(ids, C.c_int32),
                                              P(emb, C.c_double), C.byref(n_out))
    assert rc == 0
    return ids[:n_out.value].tolist(), emb[:n_out.value].copy()
