#!/usr/bin/env python3
"""Per-node graph-reconstruction AP (metrics.computeMAP semantics: node i ranks the candidates j > i with score > 0, stable descending) of a SAVED
embedding for a given node list, on the CPU, without the n x n matrix -- for the sequential oracle's embeddings, which are too large to travel to the
GPU box (scripts/make_golden_n2v_scale.py --save-emb).  Scores are fp64 dot products of the fp32 rows, ties broken by node id: the arithmetic of
gem_amd/csrc/eval.hip, so the APs pair with sampled_ap_gpu's node by node.

    python scripts/score_oracle_ap.py --emb .refruns/oracle_rmat17_f27.npy --rmat-scale 17 --edges 2000000 --seed 20260928 \
        --nodes gpurun_out/r05_rmat17/nodes_big.npy --out .refruns/ap_oracle_rmat17_f27_big.npy [--procs 4]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def ap_of_nodes(X, src_sorted_dst, starts, nodes, batch=64):
    n = X.shape[0]
    X64 = X.astype(np.float64)
    out = np.zeros(len(nodes))
    for b0 in range(0, len(nodes), batch):
        nb = nodes[b0:b0 + batch]
        S = X64 @ X64[nb].T                       # n x batch
        for k, i in enumerate(nb):
            s = S[i + 1:, k]
            nbr = src_sorted_dst[starts[i]:starts[i + 1]]
            nbr = nbr[nbr > i] - (i + 1)
            if nbr.size == 0:
                continue
            st = s[nbr]
            keep = st > 0
            nbr, st = nbr[keep], st[keep]
            if nbr.size == 0:
                continue
            pos = s[s > 0]
            # rank_all(t) = 1 + #{j: s_j > s_t} + #{j < t: s_j == s_t};  rank_hit(t) likewise among the true neighbours
            srt = np.sort(pos)
            greater = pos.size - np.searchsorted(srt, st, side='right')
            ties = np.array([int(np.count_nonzero(s[:t] == v)) for t, v in zip(nbr, st)]) if np.any(np.searchsorted(srt, st, side='right') - np.searchsorted(srt, st, side='left') > 1) else 0
            rank_all = 1 + greater + ties
            o = np.lexsort((nbr, -st))
            rank_hit = np.empty(nbr.size); rank_hit[o] = np.arange(1, nbr.size + 1)
            out[b0 + k] = float(np.mean(rank_hit / rank_all))
    return out


def _work(args):
    emb, graph_args, nodes = args
    X = np.load(emb, mmap_mode='r')
    X = np.ascontiguousarray(X)
    g = make_graph(*graph_args)
    order = np.argsort(g.src, kind='stable')
    d_sorted = g.dst[order].astype(np.int64)
    starts = np.searchsorted(g.src[order], np.arange(g.n + 1))
    return ap_of_nodes(X, d_sorted, starts, nodes)


def make_graph(kind, scale_or_nodes, edges, blocks, seed):
    from gem_amd.graph import rmat_graph, sbm_graph
    return rmat_graph(scale_or_nodes, edges, seed) if kind == 'rmat' else sbm_graph(scale_or_nodes, edges, blocks, seed)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--emb', required=True)
    ap.add_argument('--rmat-scale', type=int, default=0)
    ap.add_argument('--sbm-nodes', type=int, default=0)
    ap.add_argument('--blocks', type=int, default=1)
    ap.add_argument('--edges', type=int, required=True)
    ap.add_argument('--seed', type=int, required=True)
    ap.add_argument('--nodes', required=True, help='.npy of node ids')
    ap.add_argument('--out', required=True)
    ap.add_argument('--procs', type=int, default=2)
    a = ap.parse_args()
    nodes = np.load(a.nodes).astype(np.int64)
    gargs = ('rmat', a.rmat_scale, a.edges, 1, a.seed) if a.rmat_scale else ('sbm', a.sbm_nodes, a.edges, a.blocks, a.seed)
    t = time.time()
    chunks = [c for c in np.array_split(nodes, a.procs) if len(c)]
    if a.procs > 1:
        import multiprocessing as mp
        with mp.Pool(a.procs) as pool:
            parts = pool.map(_work, [(a.emb, gargs, c) for c in chunks])
    else:
        parts = [_work((a.emb, gargs, c)) for c in chunks]
    res = np.concatenate(parts)
    np.save(a.out, res.astype(np.float64))
    print('%s: %d nodes, MAP %.6f (%.0f s)' % (a.out, len(res), res.mean(), time.time() - t))
