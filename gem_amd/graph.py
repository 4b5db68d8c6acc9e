"""Graph ingest for the HIP backend: networkx / edge arrays -> the wire format
of the C ABI (src[], dst[], w[] in the reference's edge-iteration order), plus
seeded synthetic generators (SBM, R-MAT) that never go through networkx.

Reference wire formats being replaced: gem/utils/graph_util.py:129-140
(saveGraphToEdgeListTxt / saveGraphToEdgeListTxtn2v write exactly these triples
as text for the `gf` and `node2vec` executables).
"""
import numpy as np


class EdgeListGraph(object):
    """Minimal array-backed directed graph accepted by every learn_embedding().

    Zero-copy fast path for large graphs: holds node count and
    the directed edge triples in iteration order.  Quacks enough like the
    nx.DiGraph subset GEM's hot path touches: truthiness, len(g.nodes),
    g.number_of_nodes(), g.number_of_edges(), g.edges(data='weight').
    """

    def __init__(self, n, src, dst, w=None):
        self.n = int(n)
        self.src = np.ascontiguousarray(src, dtype=np.int32)
        self.dst = np.ascontiguousarray(dst, dtype=np.int32)
        self.w = None if w is None else np.ascontiguousarray(w, dtype=np.float32)
        if self.src.shape != self.dst.shape or self.src.ndim != 1:
            raise ValueError('src/dst must be 1-D arrays of equal length')
        if self.w is not None and self.w.shape != self.src.shape:
            raise ValueError('w must match src/dst')

    def __bool__(self):
        return self.n > 0

    __nonzero__ = __bool__

    def __len__(self):
        return self.n

    @property
    def nodes(self):
        return range(self.n)

    def number_of_nodes(self):
        return self.n

    def number_of_edges(self):
        return int(self.src.shape[0])

    def edges(self, data=None, default=1):
        w = self.w
        for k in range(self.src.shape[0]):
            if data is None:
                yield int(self.src[k]), int(self.dst[k])
            else:
                yield int(self.src[k]), int(self.dst[k]), (float(w[k]) if w is not None else default)

    def has_edge(self, i, j):
        return self._keyset_contains(i, j)

    def _keyset_contains(self, i, j):
        if not hasattr(self, '_keys'):
            self._keys = set((self.src.astype(np.int64) * self.n + self.dst).tolist())
        return (int(i) * self.n + int(j)) in self._keys

    def to_networkx(self):
        import networkx as nx
        G = nx.DiGraph()
        G.add_nodes_from(range(self.n))
        if self.w is None:
            G.add_edges_from(zip(self.src.tolist(), self.dst.tolist()), weight=1.0)
        else:
            G.add_weighted_edges_from(zip(self.src.tolist(), self.dst.tolist(), self.w.tolist()))
        return G


def edge_arrays(graph):
    """(n, src, dst, w, node_order) in the order the reference iterates.

    * n = len(graph.nodes)                       (gf.py:91, hope.py:29)
    * triples in graph.edges(data='weight', default=1) order      (gf.py:94, graph_util.py:133)
    * node_order = list(graph.nodes) -- HOPE's row order (hope.py:28 uses nx.to_numpy_matrix,
      whose rows follow insertion order, not node id).
    Node ids must be ints in [0, n) as the reference assumes (X[i] is indexed by node id).
    """
    if isinstance(graph, EdgeListGraph):
        return graph.n, graph.src, graph.dst, graph.w, None
    n = len(graph.nodes)
    m = graph.number_of_edges()
    src = np.empty(m, dtype=np.int32)
    dst = np.empty(m, dtype=np.int32)
    w = np.empty(m, dtype=np.float32)
    k = 0
    for i, j, wt in graph.edges(data='weight', default=1):
        src[k] = i
        dst[k] = j
        w[k] = wt
        k += 1
    order = np.fromiter(graph.nodes, dtype=np.int64, count=n)
    if m and (min(src.min(), dst.min()) < 0 or max(src.max(), dst.max()) >= n):
        raise ValueError('node ids must be integers in [0, n): GEM indexes the embedding by node id')
    if np.array_equal(order, np.arange(n)):
        order = None
    return n, src, dst, w, order


def to_csr(n, src, dst, w=None, sort_cols=False):
    """CSR by source (row_ptr int64, col int32, w float32), stable: keeps the
    within-row order of the input unless sort_cols."""
    src = np.asarray(src)
    dst = np.asarray(dst)
    if sort_cols:
        perm = np.lexsort((dst, src))
    else:
        perm = np.argsort(src, kind='stable')
    counts = np.bincount(src, minlength=n)
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=row_ptr[1:])
    col = np.ascontiguousarray(dst[perm], dtype=np.int32)
    ww = None if w is None else np.ascontiguousarray(np.asarray(w)[perm], dtype=np.float32)
    return row_ptr, col, ww


# ------------------------------------------------------------------ generators
def sbm_graph(n, n_directed_edges, n_blocks, seed, intra=0.8):
    """Seeded stochastic-block-model graph built directly as arrays.

    Undirected simple graph stored in both directions; `n_blocks` equal blocks of
    contiguous ids; ~`intra` of the edges inside a block; unit weights; edges
    sorted by (src, dst) so iteration order == ascending node order.
    The number of directed edges is `n_directed_edges` up to de-duplication
    (a few 1e-4 relative at the BASELINE densities).
    """
    rng = np.random.default_rng(seed)
    m_und = n_directed_edges // 2
    bs = n // n_blocks
    if bs < 2:
        raise ValueError('blocks too small')
    n_in = int(round(m_und * intra))
    n_out = m_und - n_in
    blk = rng.integers(0, n_blocks, size=n_in, dtype=np.int64)
    a_in = blk * bs + rng.integers(0, bs, size=n_in, dtype=np.int64)
    b_in = blk * bs + rng.integers(0, bs, size=n_in, dtype=np.int64)
    a_out = rng.integers(0, n_blocks * bs, size=n_out, dtype=np.int64)
    b_out = rng.integers(0, n_blocks * bs, size=n_out, dtype=np.int64)
    same = (a_out // bs) == (b_out // bs)
    # push same-block "inter" pairs to the next block so the intra fraction is as requested
    b_out[same] = (b_out[same] + bs) % (n_blocks * bs)
    a = np.concatenate([a_in, a_out])
    b = np.concatenate([b_in, b_out])
    keep = a != b
    a, b = a[keep], b[keep]
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    key = np.unique(lo * n + hi)
    lo, hi = key // n, key % n
    src = np.concatenate([lo, hi])
    dst = np.concatenate([hi, lo])
    perm = np.lexsort((dst, src))
    return EdgeListGraph(n, src[perm].astype(np.int32), dst[perm].astype(np.int32), None)


def orient_randomly(g, seed):
    """A DIRECTED graph from an undirected one stored in both directions: every undirected edge {a, b} is kept in ONE direction chosen
    by a fair coin (half the arcs).  HOPE's general case (hope.py:28-36 makes no symmetry assumption): A != A^T, S has distinct left
    and right singular vectors."""
    rng = np.random.default_rng(seed)
    src, dst = np.asarray(g.src), np.asarray(g.dst)
    und = src < dst
    a, b = src[und], dst[und]
    flip = rng.random(len(a)) < 0.5
    s2 = np.where(flip, b, a); d2 = np.where(flip, a, b)
    perm = np.lexsort((d2, s2))
    return EdgeListGraph(g.n, s2[perm].astype(np.int32), d2[perm].astype(np.int32), None)


def rmat_graph(scale, n_directed_edges, seed, a=0.57, b=0.19, c=0.19):
    """R-MAT (Chakrabarti et al.) power-law graph, symmetrised, self-loops and
    duplicates dropped; 2**scale nodes (BASELINE configs[4])."""
    rng = np.random.default_rng(seed)
    n = 1 << scale
    m = n_directed_edges // 2
    src = np.zeros(m, dtype=np.int64)
    dst = np.zeros(m, dtype=np.int64)
    for _ in range(scale):
        r = rng.random(m)
        src = (src << 1) | (r >= a + b)
        dst = (dst << 1) | (((r >= a) & (r < a + b)) | (r >= a + b + c))
    keep = src != dst
    lo, hi = np.minimum(src[keep], dst[keep]), np.maximum(src[keep], dst[keep])
    key = np.unique(lo * n + hi)
    lo, hi = key // n, key % n
    s = np.concatenate([lo, hi])
    d = np.concatenate([hi, lo])
    perm = np.lexsort((d, s))
    return EdgeListGraph(n, s[perm].astype(np.int32), d[perm].astype(np.int32), None)


def group_edges_by_source(src, dst, w=None):
    """Reorder an edge list so that every source's edges are contiguous, sources in order of first appearance, each source's edges
    in their original relative order.  For graph.edges() / saveGraphToEdgeListTxt output this is the identity.  For an interleaved
    hand-written edge file it CHANGES the visiting order of gf.cpp:157-163 / gf.py:95-100 (the reference trains such files in file
    order); GraphFactorization(..., regroup_edges=True) opts into it when libgem_hip.so rejects the file order as unschedulable."""
    src = np.asarray(src)
    first = np.full(int(src.max()) + 1 if len(src) else 0, len(src), dtype=np.int64)
    np.minimum.at(first, src, np.arange(len(src)))
    order = np.argsort(first[src], kind='stable')
    return src[order], np.asarray(dst)[order], (None if w is None else np.asarray(w)[order])
