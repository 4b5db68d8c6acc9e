#!/usr/bin/env python3
"""scripts/score_oracle_ap.py for graphs where a sort of all candidates per node is out of reach (R-MAT scale 22: 4.2 M candidates x 131 072 nodes = 13
core-hours of sorting; the 2 GB embedding cannot travel to the GPU box either: snapshots are capped at 512 MiB).  Same semantics (metrics.computeMAP: node i
ranks the candidates j > i with score > 0, stable descending; fp64 dot products of the fp32 rows; ties by node id), but the rank of a true neighbour is COUNTED
-- 1 + #{j > i: s_j > s_t} + #{i < j < t: s_j == s_t} -- one vectorised comparison per neighbour, and only nodes with many ranked neighbours sort.  Batched
fp64 GEMMs over the rows above the batch's smallest node; worker processes over node chunks.  Equal to score_oracle_ap.ap_of_nodes node by node (--selftest).

    python scripts/score_oracle_ap_counting.py --emb .refruns/oracle_rmat22_f27.npy --scale 22 --flags 27 --procs 4
writes tests/golden/n2v_ref_oracle_rmat22[_vocab_order]_e128k.json (params / engine / seconds from <emb>.json)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import numpy as np


def ap_counting(X64, d_sorted, starts, nodes, batch=64, sort_from=48):
    out = np.zeros(len(nodes))
    for b0 in range(0, len(nodes), batch):
        nb = nodes[b0:b0 + batch]
        lo = int(nb.min()) + 1
        S = X64[lo:] @ X64[nb].T                                  # rows lo.. x batch
        for k, i in enumerate(nb):
            s = S[i + 1 - lo:, k]                                  # scores of the candidates j > i (index j - i - 1)
            nbr = d_sorted[starts[i]:starts[i + 1]]
            nbr = nbr[nbr > i] - (i + 1)
            if nbr.size == 0:
                continue
            st = s[nbr]
            keep = st > 0
            nbr, st = nbr[keep], st[keep]
            if nbr.size == 0:
                continue
            if nbr.size >= sort_from:
                pos = s[s > 0]
                srt = np.sort(pos)
                greater = pos.size - np.searchsorted(srt, st, side='right')
            else:
                greater = np.array([int(np.count_nonzero(s > v)) for v in st])
            ties = np.array([int(np.count_nonzero(s[:t] == v)) for t, v in zip(nbr, st)])
            rank_all = 1 + greater + ties
            o = np.lexsort((nbr, -st))
            rank_hit = np.empty(nbr.size); rank_hit[o] = np.arange(1, nbr.size + 1)
            out[b0 + k] = float(np.mean(rank_hit / rank_all))
    return out


def _work(args):
    emb, scale, edges, nodes = args
    from gem_amd.graph import rmat_graph
    g = rmat_graph(scale, edges, 20260928)
    order = np.argsort(g.src, kind='stable')
    d_sorted = g.dst[order].astype(np.int64)
    starts = np.searchsorted(g.src[order], np.arange(g.n + 1))
    del order
    X64 = np.load(emb, mmap_mode='r').astype(np.float64)
    t = time.time()
    out = ap_counting(X64, d_sorted, starts, nodes)
    print('chunk of %d nodes in %.0f s' % (len(nodes), time.time() - t), flush=True)
    return out


if __name__ == '__main__':
    ap_ = argparse.ArgumentParser()
    ap_.add_argument('--emb'); ap_.add_argument('--scale', type=int, default=22); ap_.add_argument('--flags', type=int, default=27)
    ap_.add_argument('--sample', type=int, default=131072); ap_.add_argument('--procs', type=int, default=4); ap_.add_argument('--selftest', action='store_true')
    a = ap_.parse_args()
    from gem_amd.graph import rmat_graph
    from gem_amd.evaluation import reconstruction as gr
    if a.selftest:
        import score_oracle_ap
        g = rmat_graph(13, 160000, 20260928)
        X = np.random.RandomState(0).randn(g.n, 16).astype(np.float32); X[5] = X[7]; X[100:103] = 0
        nodes = gr.eligible_sample(g, 1500)
        order = np.argsort(g.src, kind='stable'); ds = g.dst[order].astype(np.int64); st = np.searchsorted(g.src[order], np.arange(g.n + 1))
        a1 = score_oracle_ap.ap_of_nodes(X, ds, st, np.asarray(nodes, dtype=np.int64))
        a2 = ap_counting(X.astype(np.float64), ds, st, np.asarray(nodes, dtype=np.int64), batch=7, sort_from=5)
        a3 = ap_counting(X.astype(np.float64), ds, st, np.asarray(nodes, dtype=np.int64), batch=64, sort_from=10 ** 9)
        assert np.array_equal(a1, a2) and np.array_equal(a1, a3), (np.abs(a1 - a2).max(), np.abs(a1 - a3).max())
        print('selftest ok: counting == sorting, node by node, incl. duplicated and zero rows')
        sys.exit(0)
    edges = {22: 64000000, 20: 16000000, 17: 2000000}[a.scale]
    side = json.load(open(a.emb + '.json'))
    assert side['params']['flags'] == a.flags and side['params']['rmat_scale'] == a.scale
    g = rmat_graph(a.scale, edges, 20260928)
    nodes = np.asarray(gr.eligible_sample(g, a.sample), dtype=np.int64)
    # interleave the chunks so that every worker gets low and high node ids (low ids have more candidates: more work)
    chunks = [nodes[k::a.procs] for k in range(a.procs)]
    os.environ.setdefault('OPENBLAS_NUM_THREADS', str(max(1, 8 // a.procs))); os.environ.setdefault('OMP_NUM_THREADS', str(max(1, 8 // a.procs)))
    import multiprocessing as mp
    t = time.time()
    with mp.get_context('spawn').Pool(a.procs) as pool:
        parts = pool.map(_work, [(a.emb, a.scale, edges, c) for c in chunks])
    aps = np.zeros(len(nodes))
    for k, p in enumerate(parts):
        aps[k::a.procs] = p
    name = 'n2v_ref_oracle_rmat%d%s_e%dk.json' % (a.scale, '_vocab_order' if a.flags & 16 else '', len(nodes) // 1024)
    out = {'params': side['params'], 'engine': side['engine'], 'seconds': side['seconds'], 'edges_per_s': g.number_of_edges() / side['seconds'],
           'sample': 'gem_amd.evaluation.reconstruction.eligible_sample(g, %d)  [scored by scripts/score_oracle_ap_counting.py from the saved embedding: the CPU scorer with '
                     'counted ranks, node by node equal to scripts/score_oracle_ap.py]' % len(nodes),
           'MAP': float(aps.mean()), 'MAP_se': float(aps.std(ddof=1) / np.sqrt(len(aps))), 'ap': [round(float(v), 6) for v in aps]}
    json.dump(out, open(os.path.join(ROOT, 'tests', 'golden', name), 'w'))
    print(name, 'MAP %.6f +- %.6f, scored in %.0f s' % (out['MAP'], out['MAP_se'], time.time() - t), flush=True)
