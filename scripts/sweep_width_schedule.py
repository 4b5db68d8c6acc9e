#!/usr/bin/env python3
"""Round 6: WHEN does concurrency cost quality on a power-law graph?  One node2vec pass on R-MAT scale 17 / 20 through the staged C ABI, the SGNS launch cut
into consecutive walk ranges with their own wavefront cap (gemhip_n2v_set_max_waves): schedule `f1:W1,f2:W2,...` trains walks [0, f1 n) at W1 wavefronts,
[f1 n, f2 n) at W2, ... (alpha follows the global token index as in the one-launch pass).  Each run is paired per node with the sequential oracle's APs
(tests/golden/n2v_ref_oracle_rmat{17,20}_vocab_order_e{16k,128k}.json); --save-ap keeps the per-node APs for offline analysis.

    python scripts/sweep_width_schedule.py --scale 17 --schedules '1:768;0.05:64,1:768;0.9:768,1:64' --out gpurun_out/r06c/sched17.jsonl
"""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from gem_amd import _hip
from gem_amd.graph import edge_arrays, rmat_graph
from gem_amd.evaluation import reconstruction as gr

ap_ = argparse.ArgumentParser()
ap_.add_argument('--scale', type=int, default=17)
ap_.add_argument('--schedules', default='1:768', help="';'-separated schedules of ','-separated segments frac:W[:hot-row token threshold]; W = 0: the planner's width")
ap_.add_argument('--repeats', type=int, default=1)
ap_.add_argument('--flags', type=int, default=27)
ap_.add_argument('--fresh', type=int, default=0)
ap_.add_argument('--out', default='gpurun_out/r06_sched.jsonl')
ap_.add_argument('--save-ap', default=None)
ap_.add_argument('--canaries', type=int, default=0, help='for up to this many sampled nodes whose oracle AP is 1 and whose AP here is <= 0.5: who outranks the true neighbour?')
a = ap_.parse_args()
SEED = 20260923

gpath = os.path.join(ROOT, 'tests', 'golden', 'n2v_ref_oracle_rmat%d%s_%s.json' % (a.scale, '' if a.flags == 11 else '_vocab_order', 'e16k' if a.scale == 17 else 'e128k'))
if os.path.exists(gpath):
    ref = json.load(open(gpath))
    pr = ref['params']
    nsample = len(ref['ap'])
else:           # no oracle run committed (yet) for this scale: keep the per-node APs (--save-ap) and pair them when it exists (scripts/pair_saved_aps.py)
    ref = None
    pr = dict(rmat_scale=a.scale, edges={22: 64000000, 20: 16000000, 17: 2000000}[a.scale], seed=20260928, d=128, walk_len=80, num_walks=10, window=10)
    nsample = 131072
g = rmat_graph(pr['rmat_scale'], pr['edges'], pr['seed'])
nodes = gr.eligible_sample(g, nsample)
n, src, dst, w, _ = edge_arrays(g)
from test_n2v_gpu import Dev
dev = Dev(n, src, dst, w)
L = dev.L
m = C.c_int64(); _hip.check(L.gemhip_n2v_start_nodes(dev.h, C.byref(m)))
nw = m.value * pr['num_walks']
_hip.check(L.gemhip_n2v_walks(dev.h, 1.0, 1.0, pr['num_walks'], pr['walk_len'], SEED, a.flags, 0, nw, None))
_hip.check(L.gemhip_n2v_vocab(dev.h, None))
cnt = np.empty(n, np.int32)
_hip.check(L.gemhip_n2v_build_unigram(dev.h, _hip.ptr(cnt, C.c_int32), None, None))
if a.flags & 16:
    _hip.check(L.gemhip_n2v_build_unigram_vocab_order(dev.h, a.flags, None, None, None, None))
_hip.check(L.gemhip_sgns_set_fresh(dev.h, a.fresh))
tot = nw * pr['walk_len']
os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
if a.save_ap:
    os.makedirs(a.save_ap, exist_ok=True)
    if a.scale <= 20: np.save(os.path.join(a.save_ap, 'counts_scale%d.npy' % a.scale), cnt)
    np.save(os.path.join(a.save_ap, 'nodes_scale%d.npy' % a.scale), nodes.astype(np.int32))
log = open(a.out, 'a')
P = np.empty((n, pr['d']), np.float32)
for si, sched in enumerate(a.schedules.split(';')):
    segs = [(float(s.split(':')[0]), int(s.split(':')[1]), int(s.split(':')[2]) if s.count(':') > 1 else -1) for s in sched.split(',')]      # frac:W[:hot-row token threshold]
    for rep in range(a.repeats):
        _hip.check(L.gemhip_sgns_init(dev.h, pr['d'], SEED, None, None))
        lo, secs, used = 0, [], []
        for frac, W, hot_c in segs:
            hi = min(nw, int(round(frac * nw)))
            if hi <= lo:
                continue
            _hip.check(L.gemhip_n2v_set_max_waves(dev.h, W))
            _hip.check(L.gemhip_sgns_set_hot_rows(dev.h, hot_c))
            _hip.check(L.gemhip_synchronize(None))
            t = time.time()
            _hip.check(L.gemhip_sgns_train(dev.h, pr['window'], 5, 0.025, 1, 0, lo, hi, tot, 0, SEED, a.flags, None))
            _hip.check(L.gemhip_synchronize(None))
            secs.append(round(time.time() - t, 3))
            k, wv, hot, fr = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
            _hip.check(L.gemhip_sgns_last_launch(dev.h, C.byref(k), C.byref(wv), C.byref(hot), C.byref(fr)))
            used.append([wv.value, hot.value])
            lo = hi
        _hip.check(L.gemhip_sgns_get_tables(dev.h, _hip.ptr(P, C.c_float), None))
        apv = gr.sampled_ap_gpu(g, None, P, nodes)
        rec = {'scale': a.scale, 'schedule': sched, 'rep': rep, 'flags': a.flags, 'fresh': a.fresh, 'segment_seconds': secs, 'sgns_s': round(sum(secs), 3),
               'waves_and_hot_threshold': used, 'MAP': float(apv.mean()), 'nodes': int(len(apv))}
        if ref is not None:
            dd = apv - np.asarray(ref['ap'])
            rec.update({'oracle_MAP': ref['MAP'], 'gap_pct': float(100 * dd.mean() / ref['MAP']), 'gap_se_pct': float(100 * dd.std(ddof=1) / np.sqrt(len(dd)) / ref['MAP'])})
        s = json.dumps(rec)
        print(s, flush=True); log.write(s + '\n'); log.flush()
        if a.canaries and ref is not None:
            refap = np.asarray(ref['ap'])
            bad = np.nonzero((refap >= 0.99) & (apv <= 0.5))[0][:a.canaries]
            order = np.argsort(src, kind='stable'); rp = np.searchsorted(src[order], np.arange(n + 1)); nb = dst[order]
            deg = np.diff(rp)
            for b in bad:
                i = int(nodes[b])
                sc = P[i + 1:] @ P[i]
                top = np.argsort(-sc)[:4] + i + 1
                nbrs = [int(v) for v in nb[rp[i]:rp[i + 1]] if v > i]
                crec = {'canary': i, 'count': int(cnt[i]), 'deg': int(deg[i]), 'ap_here': float(apv[b]), 'norm': float(np.linalg.norm(P[i])),
                        'true_neighbours_above': [{'id': v, 'count': int(cnt[v]), 'deg': int(deg[v]), 'score': float(P[v] @ P[i]), 'norm': float(np.linalg.norm(P[v]))} for v in nbrs[:3]],
                        'top_ranked': [{'id': int(v), 'count': int(cnt[v]), 'deg': int(deg[v]), 'score': float(P[v] @ P[i]), 'norm': float(np.linalg.norm(P[v])),
                                        'cos': float(P[v] @ P[i] / (np.linalg.norm(P[v]) * np.linalg.norm(P[i]) + 1e-30)),
                                        'neighbours': [int(u) for u in nb[rp[v]:rp[v + 1]][:4]]} for v in top]}
                s2 = json.dumps(crec); print(s2, flush=True); log.write(s2 + '\n')
            log.flush()
        if a.save_ap:
            np.save(os.path.join(a.save_ap, 'ap_scale%d_f%d_s%d_r%d.npy' % (a.scale, a.flags, si, rep)), apv.astype(np.float32))
            if a.scale <= 20: np.save(os.path.join(a.save_ap, 'norms_scale%d_f%d_s%d_r%d.npy' % (a.scale, a.flags, si, rep)), np.linalg.norm(P, axis=1).astype(np.float32))
dev.close()
