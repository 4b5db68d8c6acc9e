#!/usr/bin/env python3
"""How much of the sequential oracle's graph-reconstruction MAP on a power-law graph is the REALISATION (training seed, unigram-table layout = which node a
given random draw names as the negative) and how much is the algorithm?  Background: the two sequential oracle runs of R-MAT scale 20 differ by 18.9 %
between the layouts (flags 11 against 27; scale 17: 8.7 %), node-paired s.e. 1.3 % -- and a Hogwild launch is paired against ONE of them.  The GPU launch
draws the same negatives as the oracle run of its layout (same seed, same table), so a paired gap cancels that part; this study measures how large it is.

R-MAT scale 14 (16 384 nodes, 250 000 edges, d = 128, r = 10, l = 80, k = 10, one epoch): `--seeds` training seeds x 2 layouts, each a sequential oracle run
(about 7 min of one core), AP over every eligible node.  CPU only (test infrastructure: uses oracle/).

    python scripts/study_layout_vs_seed_oracle.py --procs 8 --out profiles/r05_oracle_layout_vs_seed_rmat14.json
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import numpy as np


def run(args):
    scale, edges, gseed, seed, flags = args
    import oracle
    from gem_amd.graph import edge_arrays, rmat_graph
    from score_oracle_ap import ap_of_nodes
    g = rmat_graph(scale, edges, gseed)
    n, src, dst, _, _ = edge_arrays(g)
    t = time.time()
    X, _ = oracle.n2v_train(n, src, dst, None, 128, 80, 10, 10, 1, 1.0, 1.0, seed, flags)
    el = time.time() - t
    order = np.argsort(src, kind='stable')
    starts = np.searchsorted(src[order], np.arange(n + 1))
    nodes = np.unique(src[dst > src]).astype(np.int64)
    ap = ap_of_nodes(np.asarray(X, np.float32), dst[order].astype(np.int64), starts, nodes)
    return seed, flags, el, ap


if __name__ == '__main__':
    p = argparse.ArgumentParser()
    p.add_argument('--scale', type=int, default=14); p.add_argument('--edges', type=int, default=250000)
    p.add_argument('--graph-seed', type=int, default=20260928)
    p.add_argument('--seeds', default='20260923,1,2,3')
    p.add_argument('--procs', type=int, default=8)
    p.add_argument('--out', required=True)
    a = p.parse_args()
    seeds = [int(s) for s in a.seeds.split(',')]
    jobs = [(a.scale, a.edges, a.graph_seed, s, fl) for s in seeds for fl in (11, 27)]
    import multiprocessing as mp
    with mp.Pool(a.procs) as pool:
        res = pool.map(run, jobs, chunksize=1)
    aps = {(s, fl): ap for s, fl, _, ap in res}
    rec = {'graph': 'R-MAT scale %d, %d edges requested, graph seed %d' % (a.scale, a.edges, a.graph_seed), 'eligible_nodes': int(len(res[0][3])),
           'runs': [{'seed': s, 'flags': fl, 'seconds': round(el, 1), 'MAP': float(ap.mean())} for s, fl, el, ap in res]}

    def paired(x, y):
        d = x - y
        return {'diff_pct': float(100 * d.mean() / y.mean()), 'node_se_pct': float(100 * d.std(ddof=1) / np.sqrt(len(d)) / y.mean()), 'corr': float(np.corrcoef(x, y)[0, 1])}
    rec['layout_11_vs_27_same_seed'] = [dict(seed=s, **paired(aps[(s, 11)], aps[(s, 27)])) for s in seeds]
    rec['seed_vs_first_seed_same_layout'] = [dict(seed=s, flags=fl, **paired(aps[(s, fl)], aps[(seeds[0], fl)])) for fl in (11, 27) for s in seeds[1:]]
    for fl in (11, 27):
        m = np.array([aps[(s, fl)].mean() for s in seeds])
        rec['MAP_over_seeds_flags%d' % fl] = {'mean': float(m.mean()), 'sd_pct': float(100 * m.std(ddof=1) / m.mean())}
    json.dump(rec, open(a.out, 'w'), indent=1)
    print(json.dumps(rec, indent=1))
