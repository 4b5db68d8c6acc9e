"""Train/test edge split for link prediction -- array restatement of
gem/utils/evaluation_util.py:39-53 (split_di_graph_to_train_test): every edge (for undirected graphs: every pair
st < ed, both directions together) stays in the TRAIN graph with probability train_ratio (np.random.uniform() <=
train_ratio) and goes to the TEST graph otherwise.  Works on EdgeListGraph at any size; nx graphs are converted.
The draw order follows the reference (one uniform per candidate edge in graph.edges order), so np.random.seed
reproduces the reference's split."""
import numpy as np

from gem_amd.graph import EdgeListGraph, edge_arrays


def split_di_graph_to_train_test(di_graph, train_ratio, is_undirected=True):
    n, src, dst, w, _ = edge_arrays(di_graph)
    m = len(src)
    if is_undirected:
        cand = src < dst                               # evaluation_util.py:44: `if is_undirected and st >= ed: continue`
    else:
        cand = np.ones(m, dtype=bool)
    u = np.random.uniform(size=int(cand.sum()))       # one draw per candidate edge, in edge order (:46)
    keep_train = np.zeros(m, dtype=bool)
    idx = np.flatnonzero(cand)
    keep_train[idx] = u <= train_ratio
    if is_undirected:
        # the reverse direction follows its forward edge (:48-49, :52-53)
        key_fwd = src[idx].astype(np.int64) * n + dst[idx]
        order = np.argsort(key_fwd)
        rev = np.flatnonzero(~cand & (src != dst))
        key_rev = dst[rev].astype(np.int64) * n + src[rev]
        pos = np.searchsorted(key_fwd[order], key_rev)
        ok = (pos < len(order)) & (key_fwd[order][np.minimum(pos, len(order) - 1)] == key_rev)
        keep_train[rev[ok]] = keep_train[idx[order[pos[ok]]]]
        keep_train[rev[~ok]] = True                    # a lone reverse edge is never visited by the reference: stays in both graphs
        lone = np.zeros(m, dtype=bool); lone[rev[~ok]] = True
        loops = src == dst
        keep_train[loops] = True; lone |= loops
    else:
        lone = np.zeros(m, dtype=bool)
    in_test = ~keep_train | lone
    ww = None if w is None else w
    train = EdgeListGraph(n, src[keep_train], dst[keep_train], None if ww is None else ww[keep_train])
    test = EdgeListGraph(n, src[in_test], dst[in_test], None if ww is None else ww[in_test])
    return train, test
