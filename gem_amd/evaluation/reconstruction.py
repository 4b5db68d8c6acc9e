"""Graph-reconstruction metric (MAP, precision curve) -- vectorised restatement of
GEM's evaluator, used as THE parity metric of this backend.

Mirrors, result-for-result (pinned by tests/golden/map_ref.json, which was
produced by the reference's own code):
  gem/evaluation/evaluate_graph_reconstruction.py:8-46   evaluateStaticGraphReconstruction
  gem/utils/evaluation_util.py:20-36                     get_edge_list_from_adj_mtrx (i<j, adj>0)
  gem/evaluation/metrics.py:6-24, 27-46                  computePrecisionCurve, computeMAP

The reference builds Python lists of (i, j, w) tuples and sorts them (O(n^2) Python
objects); here every node's candidate list is one stable argsort over a numpy
row, so n ~ 2e4 is practical.  Tie order follows the reference: Python's stable
`sorted(..., reverse=True)` keeps equal weights in ascending-j order.
"""
import numpy as np


def _adjacency_bool(digraph, n):
    A = np.zeros((n, n), dtype=bool)
    if hasattr(digraph, 'src'):
        A[digraph.src, digraph.dst] = True
    else:
        for i, j in digraph.edges():
            A[i, j] = True
    return A


def average_precision_rows(score, truth, undirected=True):
    """Per-node AP exactly as metrics.computeMAP: node i ranks candidates j (j>i when
    undirected) with score>0 by descending score; AP_i = mean precision at the hits."""
    n = score.shape[0]
    ap = np.zeros(n)
    for i in range(n):
        lo = i + 1 if undirected else 0
        s = score[i, lo:]
        t = truth[i, lo:]
        if not undirected:
            keep = np.ones(s.shape[0], dtype=bool)
            keep[i] = False
            s, t = s[keep], t[keep]
        pos = s > 0
        s, t = s[pos], t[pos]
        if s.size == 0:
            continue
        order = np.argsort(-s, kind='stable')
        hit = t[order]
        nh = hit.sum()
        if nh == 0:
            continue
        prec = np.cumsum(hit) / np.arange(1, hit.size + 1)
        ap[i] = prec[hit].sum() / nh
    return ap


def precision_curve(score, truth, undirected=True, max_k=None):
    n = score.shape[0]
    iu = np.triu_indices(n, 1) if undirected else np.where(~np.eye(n, dtype=bool))
    s = score[iu]
    t = truth[iu]
    pos = s > 0
    s, t = s[pos], t[pos]
    order = np.argsort(-s, kind='stable')
    hit = t[order]
    if max_k is not None:
        hit = hit[:max_k]
    return np.cumsum(hit) / np.arange(1, hit.size + 1)


def evaluateStaticGraphReconstruction(digraph, graph_embedding, X_stat, node_l=None, file_suffix=None,
                                      sample_ratio_e=None, is_undirected=True, is_weighted=False):
    """Same signature and return tuple as the reference: (MAP, prec_curv, err, err_baseline)."""
    n = len(digraph.nodes)
    est = graph_embedding.get_reconstructed_adj(X_stat, node_l)
    truth = _adjacency_bool(digraph, n)
    if sample_ratio_e:
        raise NotImplementedError('sample_ratio_e: use sampled_map() for large graphs')
    ap = average_precision_rows(est, truth, undirected=is_undirected)
    if is_undirected:
        MAP = ap.sum() / n
    else:
        has_out = truth.any(axis=1)
        MAP = ap[has_out].sum() / max(int(has_out.sum()), 1)
    prec = precision_curve(est, truth, undirected=is_undirected)
    err = err_base = None
    if is_weighted:
        W = np.zeros((n, n))
        for i, j, w in digraph.edges(data='weight', default=1):
            W[i, j] = w
        e = est.copy()
        e[W == 0] = 0
        err = np.linalg.norm(W - e)
        err_base = np.linalg.norm(W)
    return float(MAP), prec.tolist(), err, err_base


def sampled_map(graph, pair_score, nodes, undirected=True):
    """MAP over a sample of nodes for graphs where the n x n matrix cannot be formed
    (SURVEY 8f row 1).  `pair_score(i) -> scores of node i against all nodes` (length n).
    Equals computeMAP restricted to `nodes` (tested against the full evaluator)."""
    n = graph.number_of_nodes()
    if hasattr(graph, 'src'):
        order = np.argsort(graph.src, kind='stable')
        s_sorted = graph.src[order]
        d_sorted = graph.dst[order]
        starts = np.searchsorted(s_sorted, np.arange(n + 1))
        nbrs = lambda i: d_sorted[starts[i]:starts[i + 1]]
    else:
        nbrs = lambda i: np.fromiter(graph.successors(i), dtype=np.int64)
    aps = []
    for i in nodes:
        s = np.asarray(pair_score(i), dtype=np.float64).copy()
        t = np.zeros(n, dtype=bool)
        t[nbrs(i)] = True
        lo = i + 1 if undirected else 0
        s, t = s[lo:], t[lo:]
        if not undirected:
            s[i] = 0
        pos = s > 0
        s, t = s[pos], t[pos]
        if s.size == 0 or t.sum() == 0:
            aps.append(0.0)
            continue
        hit = t[np.argsort(-s, kind='stable')]
        prec = np.cumsum(hit) / np.arange(1, hit.size + 1)
        aps.append(prec[hit].sum() / hit.sum())
    return float(np.mean(aps)) if aps else 0.0


def eligible_sample(graph, size, seed=1):
    """A node sample for sampled MAP on graphs where a uniform sample is mostly dead weight.  metrics.computeMAP ranks, for node i, the candidates j > i
    only (evaluation_util.py:28-35 lists the upper triangle): a node without a neighbour j > i has AP 0 whatever the embedding.  On a power-law graph that is
    two thirds of the nodes, the remaining APs are small and heavy-tailed, and a 2048-node uniform sample of R-MAT scale 17 sums to ~11: ONE node whose only
    neighbour lands on rank 1 moves the "MAP" by 9 %.  This draws `size` nodes (sorted; RandomState(seed), without replacement) from the nodes that have
    such a neighbour -- every sampled node carries information, and paired launch-to-launch comparisons get a standard error of ~0.3 % at 16 384 nodes."""
    from gem_amd.graph import edge_arrays
    n, src, dst, _, _ = edge_arrays(graph)
    elig = np.unique(src[dst > src])
    return np.sort(np.random.RandomState(seed).choice(elig, size=min(int(size), len(elig)), replace=False)).astype(np.int64)


def sampled_ap_gpu(graph, graph_embedding, X, nodes, is_undirected=True):
    """Per-node AP of graph reconstruction for `nodes`, computed on the GPU (gem_amd/csrc/eval.hip) with the same
    semantics as average_precision_rows / metrics.computeMAP -- no n x n matrix, so it works at 1M nodes.
    mean() of the result over ALL nodes equals evaluateStaticGraphReconstruction's MAP (tests/test_eval_gpu.py)."""
    import ctypes as C
    from gem_amd import _hip
    from gem_amd.graph import edge_arrays, to_csr
    n, src, dst, w, _ = edge_arrays(graph)
    row_ptr, col, _ = to_csr(n, src, dst, None)
    X = np.asarray(X)
    d = X.shape[1]
    name = graph_embedding.get_method_name() if graph_embedding is not None else ''
    if name == 'hope_gsvd':                                   # hope.py:43-44: X[i, :k] . X[j, k:]
        k = d // 2
        A = np.ascontiguousarray(X[:, :k], dtype=np.float32); B = np.ascontiguousarray(X[:, k:2 * k], dtype=np.float32)
    elif name in ('', 'graph_factor_sgd', 'node2vec_rw'):     # gf.py:103-104 / node2vec.py:56-57: X[i] . X[j]
        A = np.ascontiguousarray(X, dtype=np.float32); B = None
    else:                                                     # lap.py / lle.py score by exp(-||x_i - x_j||^2): not an inner product
        raise NotImplementedError('sampled_ap_gpu scores inner-product methods (GF, node2vec, HOPE); %r defines get_edge_weight '
                                  'differently -- use evaluateStaticGraphReconstruction' % name)
    nodes = np.ascontiguousarray(nodes, dtype=np.int32)
    ap = np.zeros(len(nodes))
    _hip.require_device()
    _hip.check(_hip.lib().gemhip_eval_sampled_ap(n, A.shape[1], A.shape[1], _hip.ptr(A, C.c_float), _hip.ptr(B, C.c_float),
                                                 _hip.ptr(row_ptr, C.c_int64), _hip.ptr(col, C.c_int32), 1 if is_undirected else 0,
                                                 len(nodes), _hip.ptr(nodes, C.c_int32), _hip.ptr(ap, C.c_double)))
    return ap
