#!/usr/bin/env python3
"""Launch-to-launch spread of the Hogwild SGNS default on R-MAT scale 17: N launches of node2vec.learn_embedding in ONE process per
unigram-table layout (flags 11 = node-id order, 27 = the binary's vocabulary order), then a sweep of the concurrent-wavefront cap on the staged API.

Every launch is scored on TWO node samples:
  * `old`: np.random.RandomState(0).choice(n, 2048) -- the sample of tests/golden/n2v_ref_oracle_rmat17*.json (rounds 2-4).  Only ~690 of those nodes can
    have a non-zero AP at all (the evaluator only ranks candidates j > i), the APs sum to ~11, so ONE node whose single neighbour lands on rank 1 moves
    the "MAP gap" by 9 %: a heavy-tailed statistic.
  * `big`: 16 384 nodes drawn (RandomState(1)) from the nodes that HAVE a neighbour j > i.
The per-node APs are written to --out as .npy so that they can be paired with the sequential oracle's APs on the build container
(scripts/score_rmat_oracle.py; the oracle's 67 MB embedding does not travel to the GPU box).

    python scripts/check_rmat17_launches.py --launches 6 --out gpurun_out/r05_rmat17
"""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from gem_amd import _hip
from gem_amd.graph import edge_arrays, rmat_graph
from gem_amd.evaluation import reconstruction as gr

ap_ = argparse.ArgumentParser()
ap_.add_argument('--launches', type=int, default=6)
ap_.add_argument('--layouts', default='11,27')
ap_.add_argument('--scale', type=int, default=17)
ap_.add_argument('--edges', type=int, default=2000000)
ap_.add_argument('--big', type=int, default=16384)
ap_.add_argument('--widths', default='0,1024,512,256,128')
ap_.add_argument('--out', default='gpurun_out/r05_rmat17')
ap_.add_argument('--tag', default='')
ap_.add_argument('--width-layouts', default='27')
ap_.add_argument('--width-launches', type=int, default=1)
ap_.add_argument('--hot-counts', default='', help='staged API, default width: explicit hot-row thresholds (token counts) to try')
ap_.add_argument('--hot-width', type=int, default=0, help='wavefront cap used together with --hot-counts (0 = the width the planner takes for that threshold)')
ap_.add_argument('--save-counts', action='store_true')
a = ap_.parse_args()
os.makedirs(a.out, exist_ok=True)
SEED = 20260923


def samples(g):
    n, src, dst, _, _ = edge_arrays(g)
    elig = np.unique(src[dst > src])
    old = np.random.RandomState(0).choice(g.n, size=2048, replace=False)
    big = np.sort(np.random.RandomState(1).choice(elig, size=min(a.big, len(elig)), replace=False))
    return old.astype(np.int32), big.astype(np.int32), len(elig)


g = rmat_graph(a.scale, a.edges, 20260928)
old, big, n_elig = samples(g)
np.save(os.path.join(a.out, 'nodes_old.npy'), old); np.save(os.path.join(a.out, 'nodes_big.npy'), big)
gold = {}
for fl, name in ((11, 'n2v_ref_oracle_rmat17.json'), (27, 'n2v_ref_oracle_rmat17_vocab_order.json')):        # (the old 2 048-node statistic, scale 17 only)
    p = os.path.join(ROOT, 'tests', 'golden', name)
    if a.scale == 17 and os.path.exists(p):
        gold[fl] = json.load(open(p))
log = open(os.path.join(a.out, 'launches%s.jsonl' % a.tag), 'a')


def emit(rec):
    s = json.dumps(rec)
    print(s, flush=True); log.write(s + '\n'); log.flush()


emit({'graph': 'rmat', 'scale': a.scale, 'n': g.n, 'directed_edges': g.number_of_edges(), 'eligible_nodes': int(n_elig), 'big_sample': int(len(big)),
      'host': os.uname().nodename, 'pid': os.getpid()})


def score(X, flags, rec):
    t = time.time()
    ap_old = gr.sampled_ap_gpu(g, None, X, old)
    ap_big = gr.sampled_ap_gpu(g, None, X, big)
    rec['eval_seconds'] = round(time.time() - t, 2)
    rec['MAP_old'] = float(ap_old.mean()); rec['MAP_big'] = float(ap_big.mean()); rec['max_ap_old'] = float(ap_old.max())
    if flags in gold:
        ref = gold[flags]
        dd = ap_old - np.asarray(ref['ap'])
        rec['gap_old_pct'] = float(100 * dd.mean() / ref['MAP']); rec['gap_old_se_pct'] = float(100 * dd.std(ddof=1) / np.sqrt(len(dd)) / ref['MAP'])
    return ap_old.astype(np.float32), ap_big.astype(np.float32)


from gem_amd.embedding.node2vec import node2vec
for fl in [int(v) for v in a.layouts.split(',') if v]:
    olds, bigs = [], []
    for k in range(a.launches):
        m = node2vec(d=128, max_iter=1, walk_len=80, num_walks=10, con_size=10, ret_p=1, inout_p=1, seed=SEED, flags=fl)
        t = time.time()
        X = m.learn_embedding(graph=g, is_weighted=True, no_python=True)
        rec = {'mode': 'one_shot', 'flags': fl, 'launch': k, 'wall_s': round(time.time() - t, 3), 'sgns_s': round(m._stats['sgns_seconds'], 3)}
        o, b = score(X, fl, rec)
        olds.append(o); bigs.append(b)
        emit(rec)
    if olds:
        np.save(os.path.join(a.out, 'ap_old_f%d%s.npy' % (fl, a.tag)), np.stack(olds)); np.save(os.path.join(a.out, 'ap_big_f%d%s.npy' % (fl, a.tag)), np.stack(bigs))

# wavefront-cap / hot-threshold sweeps on the staged API
widths = [int(v) for v in a.widths.split(',') if v != '']
hots = [int(v) for v in a.hot_counts.split(',') if v != '']
if widths or hots:
    from test_n2v_gpu import Dev
    n, src, dst, w, _ = edge_arrays(g)
    dev = Dev(n, src, dst, w)
    m = C.c_int64(); _hip.check(dev.L.gemhip_n2v_start_nodes(dev.h, C.byref(m)))
    nw = m.value * 10
    P = np.empty((n, 128), np.float32)
    for fl in [int(v) for v in a.width_layouts.split(',') if v]:
        _hip.check(dev.L.gemhip_n2v_walks(dev.h, 1.0, 1.0, 10, 80, SEED, fl, 0, nw, None))
        cnt = np.empty(n, np.int32)
        _hip.check(dev.L.gemhip_n2v_vocab(dev.h, None))
        if True:
            # token counts (for the planner replay): the node-id builder returns them; rebuild the vocabulary-order table afterwards if needed
            U = np.empty(n, np.float32); K = np.empty(n, np.int32)
            _hip.check(dev.L.gemhip_n2v_build_unigram(dev.h, _hip.ptr(cnt, C.c_int32), _hip.ptr(U, C.c_float), _hip.ptr(K, C.c_int32)))
            if fl & 16:
                _hip.check(dev.L.gemhip_n2v_build_unigram_vocab_order(dev.h, fl, None, None, None, None))
        if a.save_counts:
            np.save(os.path.join(a.out, 'counts_scale%d.npy' % a.scale), cnt)
        k_, w_, hot_, ne_, nec_ = C.c_int32(), C.c_int32(), C.c_int32(), C.c_double(), C.c_double()
        _hip.check(dev.L.gemhip_sgns_plan_launch(_hip.ptr(cnt, C.c_int32), n, 128, 10, 80, nw, fl, C.byref(k_), C.byref(w_), C.byref(hot_), C.byref(ne_), C.byref(nec_)))
        emit({'mode': 'plan', 'flags': fl, 'waves': w_.value, 'hot_thr': hot_.value, 'hot_rows': int((cnt >= hot_.value).sum()) if hot_.value > 0 else 0,
              'n_eff': ne_.value, 'n_eff_cold': nec_.value, 'tokens': int(cnt.sum()), 'active': int((cnt > 0).sum())})
        olds, bigs, labels = [], [], []
        for kind, vals in (('max_waves', widths), ('hot_count', hots)):
            for v in vals:
                for rep in range(a.width_launches):
                    _hip.check(dev.L.gemhip_n2v_set_max_waves(dev.h, v if kind == 'max_waves' else a.hot_width))
                    _hip.check(dev.L.gemhip_sgns_set_hot_rows(dev.h, v if kind == 'hot_count' else -1))
                    _hip.check(dev.L.gemhip_sgns_init(dev.h, 128, SEED, None, None))
                    _hip.check(dev.L.gemhip_synchronize(None))
                    t = time.time()
                    _hip.check(dev.L.gemhip_sgns_train(dev.h, 10, 5, 0.025, 1, 0, 0, nw, nw * 80, 0, SEED, fl, None))
                    _hip.check(dev.L.gemhip_synchronize(None))
                    el = time.time() - t
                    _hip.check(dev.L.gemhip_sgns_get_tables(dev.h, _hip.ptr(P, C.c_float), None))
                    rec = {'mode': 'staged', 'flags': fl, kind: v, 'rep': rep, 'sgns_s': round(el, 3)}
                    if kind == 'hot_count':
                        rec['max_waves'] = a.hot_width
                    o, b = score(P, fl, rec)
                    olds.append(o); bigs.append(b); labels.append([fl, 0 if kind == 'max_waves' else 1, v, rep])
                    emit(rec)
        if olds:
            np.save(os.path.join(a.out, 'ap_old_sweep_f%d%s.npy' % (fl, a.tag)), np.stack(olds)); np.save(os.path.join(a.out, 'ap_big_sweep_f%d%s.npy' % (fl, a.tag)), np.stack(bigs))
        np.save(os.path.join(a.out, 'sweep_labels_f%d%s.npy' % (fl, a.tag)), np.asarray(labels))
    dev.close()
