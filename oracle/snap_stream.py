"""oracle/snap_stream.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Sequential restatement of the WALK half of the reference's node2vec path -- the prebuilt SNAP binary gem/c_exe/node2vec that
gem/embedding/node2vec.py:34-48 runs -- INCLUDING ITS RANDOM STREAM, so that it can be compared with the binary walk for walk.
The binary's source is third-party and absent from /root/reference (snap-stanford/snap, examples/node2vec + snap-adv/n2v.cpp,
biasedrandomwalk.cpp; ELF banner "Apr 9 2017", not stripped); what is restated here was read off the ELF's symbols and
disassembly and then PINNED against the binary itself, made deterministic with oracle/shim/faketime.c (time() fixed) and
OMP_NUM_THREADS=1: tests/golden/n2v_snap_stream_walks.json holds walk matrices dumped from the running binary
(scripts/make_golden_n2v_snap_stream.py), tests/test_oracle_n2v.py asserts this module reproduces them bit for bit.

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


def simulate_walks(order, nbr, w, p, q, num_walks, walk_len, seed):
    """The walk matrix [num_walks * N][walk_len] (int32, zero padded) node2vec() hands to LearnEmbeddings."""
    tables = preprocess_transition_probs(order, nbr, w, float(p), float(q))
    rnd = TRnd(seed)
    ids = list(order)
    n = len(ids)
    out = np.zeros((num_walks * n, walk_len), dtype=np.int32)
    for r in range(num_walks):
        shuffle(ids, rnd)
        for j, s in enumerate(ids):
            wk = [s]
            if walk_len > 1 and nbr[s]:
                wk.append(nbr[s][rnd.uni_dev_int(len(nbr[s]))])
                while len(wk) < walk_len:
                    dst, src = wk[-1], wk[-2]
                    if not nbr[dst]:
                        break
                    wk.append(nbr[dst][alias_draw_int(tables[(src, dst)], rnd)])
            out[r * n + j, :len(wk)] = wk
    return out
