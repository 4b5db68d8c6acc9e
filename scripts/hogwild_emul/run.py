#!/usr/bin/env python3
"""Driver of scripts/hogwild_emul/hogwild_emul.c: which of sgns_win_kernel's departures from the sequential TrainModel costs MAP?

    python scripts/hogwild_emul/run.py --selftest
    python scripts/hogwild_emul/run.py --nodes 16384 --edges 163840 --blocks 16 --walks 4 --cfg seq W64:L2:ctr1:ctx2:neg1 ...

A configuration is `seq` (oracle_sgns_train) or W<waves>:L<prefetch distance>:ctr<m>:ctx<m>:neg<m>[:R<radius>] with the modes of
hogwild_emul.c (0 direct, 1 private copy + overwrite, 2 private copy + delta); `:hot1` keeps the nodes expected in another
wavefront's window (count >= tokens / ((W-1)(2R+1)); hotK: K times that many copies) out of the window caches and applies their negative updates at store time.  The kernel as shipped in round 2 is
ctr1:ctx2:neg1:L2.  Prints one JSON line per configuration: MAP over a fixed node sample (paired across configurations: same
walks, same draws), the fraction of negative-row / centre-row stores that overwrote a foreign update.
"""
import argparse, ctypes as C, json, os, subprocess, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
from gem_amd.graph import sbm_graph, rmat_graph, edge_arrays


def emul_lib():
    so = os.path.join(HERE, 'libhogwild_emul.so')
    src = os.path.join(HERE, 'hogwild_emul.c')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['gcc', '-O2', '-fPIC', '-std=c11', '-ffp-contract=off', '-w', '-shared', '-o', so, src, '-lm'])
    L = C.CDLL(so)
    f32p, i32p, i64p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    L.hogwild_emul_train.restype = None
    L.hogwild_emul_train.argtypes = [C.c_int64, C.c_int32, C.c_int64, C.c_int32, i32p, C.c_int32, C.c_float, C.c_int32, C.c_int32, C.c_int64,
                                     C.c_int64, C.c_int64, f32p, i32p, C.c_uint64, C.c_int32, f32p, f32p, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_int32, i64p, i32p, C.c_int32]
    return L


def p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def parse_cfg(s, window):
    if s == 'seq':
        return None
    o = dict(W=1, L=2, ctr=1, ctx=2, neg=1, R=window, hot=0)
    for part in s.split(':'):
        for k in ('ctr', 'ctx', 'neg', 'hot', 'W', 'L', 'R'):
            if part.startswith(k) and part[len(k):].isdigit():
                o[k] = int(part[len(k):]); break
        else:
            raise SystemExit('bad cfg part ' + part)
    return o


def sampled_aps(g, X, nodes):
    n = g.n
    Xd = X.astype(np.float64)
    order = np.argsort(g.src, kind='stable')
    s_sorted, d_sorted = g.src[order], g.dst[order]
    starts = np.searchsorted(s_sorted, np.arange(n + 1))
    aps = []
    for i in nodes:
        s = Xd @ Xd[i]
        tr = np.zeros(n, dtype=bool); tr[d_sorted[starts[i]:starts[i + 1]]] = True
        s, tr = s[i + 1:], tr[i + 1:]
        pos = s > 0
        s, tr = s[pos], tr[pos]
        if s.size == 0 or tr.sum() == 0:
            aps.append(0.0); continue
        hit = tr[np.argsort(-s, kind='stable')]
        prec = np.cumsum(hit) / np.arange(1, hit.size + 1)
        aps.append(float(prec[hit].sum() / hit.sum()))
    return np.asarray(aps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--nodes', type=int, default=16384)
    ap.add_argument('--edges', type=int, default=163840)
    ap.add_argument('--blocks', type=int, default=16)
    ap.add_argument('--rmat-scale', type=int, default=0)
    ap.add_argument('--d', type=int, default=128)
    ap.add_argument('--walks', type=int, default=10)
    ap.add_argument('--walk-len', type=int, default=80)
    ap.add_argument('--window', type=int, default=10)
    ap.add_argument('--seed', type=int, default=20260923)
    ap.add_argument('--sample', type=int, default=2048)
    ap.add_argument('--cfg', nargs='+', default=['seq'])
    ap.add_argument('--selftest', action='store_true')
    ap.add_argument('--out', default=None)
    ap.add_argument('--train-seed', type=int, default=None, help='seed of walks / init / draws when it differs from the graph seed (--seed)')
    ap.add_argument('--baseline-emb', default=None, help='.npy embedding of the sequential pass (oracle.n2v_train with the same seeds and flags): `seq` is then not trained again')
    ap.add_argument('--flags', type=int, default=11, help='11 = node-id table layout, 27 = the binary\'s (first-appearance order; round 5)')
    ap.add_argument('--eligible', action='store_true', help='score nodes that have a ranked neighbour (reconstruction.eligible_sample) instead of a uniform sample')
    a = ap.parse_args()
    L = emul_lib()
    if a.selftest:
        a.nodes, a.edges, a.blocks, a.walks, a.d = 512, 4096, 4, 2, 16
        a.cfg = ['seq', 'W1:ctr0:ctx0:neg0', 'W1:ctr1:ctx1:neg1:L2', 'W1:ctr2:ctx2:neg2:L3', 'W8:ctr0:ctx0:neg0', 'W8:ctr1:ctx2:neg1:L2']
    if a.rmat_scale:
        g = rmat_graph(a.rmat_scale, a.edges, a.seed)
    else:
        g = sbm_graph(a.nodes, a.edges, a.blocks, a.seed + 4)
    n, src, dst, w, _ = edge_arrays(g)
    row_ptr, col, ww = oracle.sorted_csr(n, src, dst, w)
    flags = a.flags
    tseed = a.seed if a.train_seed is None else a.train_seed
    walks = oracle.n2v_walks(row_ptr, col, None, None, 1.0, 1.0, a.walks, a.walk_len, tseed, flags)
    counts = np.ascontiguousarray(oracle.n2v_vocab(n, walks), dtype=np.int32)
    slot_tab = None
    if flags & 16:
        slot_tab, UT, KT = oracle.unigram_build_vocab_order(counts, walks, flags)[:3]
        slot_tab = np.ascontiguousarray(slot_tab, dtype=np.int32)
        L.hogwild_emul_set_slot_table.argtypes = [C.POINTER(C.c_int32), C.c_int64]; L.hogwild_emul_set_slot_table.restype = None
        L.hogwild_emul_set_slot_table(p(slot_tab, C.c_int32), len(slot_tab))
    else:
        UT, KT = oracle.unigram_build(counts)
    UT = np.ascontiguousarray(UT, dtype=np.float32); KT = np.ascontiguousarray(KT, dtype=np.int32)
    walks = np.ascontiguousarray(walks, dtype=np.int32)
    if a.eligible:
        from gem_amd.evaluation.reconstruction import eligible_sample
        nodes = eligible_sample(g, a.sample)
    else:
        nodes = np.random.RandomState(0).choice(n, size=min(a.sample, n), replace=False)
    base = None
    ref_X = None
    for cs in a.cfg:
        cfg = parse_cfg(cs, a.window)
        P, N = oracle.sgns_init(n, a.d, tseed)
        t = time.time()
        st = (C.c_int64 * 4)()
        if cfg is None and a.baseline_emb:
            P = np.ascontiguousarray(np.load(a.baseline_emb), dtype=np.float32)
        elif cfg is None:
            if slot_tab is not None:
                oracle.sgns_train_vocab_order(walks, a.window, 0.025, 1, 0, walks.size, 0, 0, slot_tab, UT, KT, tseed, flags, P, N)
            else:
                oracle.sgns_train(walks, a.window, 0.025, 1, 0, walks.size, 0, 0, UT, KT, tseed, flags, P, N)
        else:
            L.hogwild_emul_train(n, a.d, walks.shape[0], walks.shape[1], p(walks, C.c_int32), a.window, 0.025, 1, 0, walks.size, 0, 0,
                                 p(UT, C.c_float), p(KT, C.c_int32), tseed, flags, p(P, C.c_float), p(N, C.c_float),
                                 cfg['W'], cfg['L'], cfg['R'], cfg['ctr'], cfg['ctx'], cfg['neg'], st, p(counts, C.c_int32),
                                 0 if not cfg['hot'] else max(2, int(np.ceil(walks.size / ((cfg['W'] - 1) * (2 * cfg['R'] + 1) * cfg['hot'])))))
        el = time.time() - t
        aps = sampled_aps(g, P, nodes)
        if base is None:
            base = aps; ref_X = P.copy()
        dlt = aps - base
        rec = dict(cfg=cs, flags=flags, n=n, walks=a.walks, d=a.d, MAP=float(aps.mean()), MAP_se=float(aps.std(ddof=1) / np.sqrt(len(aps))),
                   rel_vs_first_pct=float(100 * dlt.mean() / base.mean()), rel_se_pct=float(100 * dlt.std(ddof=1) / np.sqrt(len(aps)) / base.mean()),
                   max_abs_diff_vs_first=float(np.abs(P - ref_X).max()), seconds=round(el, 1), pairs=int(st[0]),
                   neg_overwrote_foreign=float(st[1]) / max(1, 5 * st[0]), centre_overwrote_foreign=int(st[2]))
        print(json.dumps(rec), flush=True)
        if a.out:
            open(a.out, 'a').write(json.dumps(rec) + '\n')


if __name__ == '__main__':
    main()
