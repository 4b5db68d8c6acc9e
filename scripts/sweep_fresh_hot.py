#!/usr/bin/env python3
"""Round 6: what does the FRESHNESS of hot rows buy on a power-law graph?  One Hogwild launch of node2vec.learn_embedding per configuration
`W:fresh:flags[:hot_count[:neg_count]][*repeats]` (GEMHIP_SGNS_MAX_WAVES : GEMHIP_SGNS_FRESH : unigram-table layout 11 | 27 : GEMHIP_SGNS_HOT_COUNT :
GEMHIP_SGNS_NEG_COUNT; W = 0 lets the planner choose) on R-MAT scale 17 / 20, each
paired per node with the sequential oracle's APs of tests/golden/n2v_ref_oracle_rmat{17,20}*_e{16k,128k}.json (same graph, seed and node sample as
tests/test_rmat_gpu.py).  With GEM_HIP_LIB=gem_amd/libgem_hip_stale.so (scripts/build_variant.sh stale -DGEMHIP_SGNS_STALENESS) every launch also
appends its staleness histogram to $GEMHIP_SGNS_STALENESS_OUT.

    python scripts/sweep_fresh_hot.py --scale 20 --configs 207:0:27,768:0:27,768:3:27 --out gpurun_out/r06_fresh.jsonl
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

ap_ = argparse.ArgumentParser()
ap_.add_argument('--scale', type=int, default=17)
ap_.add_argument('--configs', default='256:0:27,768:0:27,768:3:27')
ap_.add_argument('--out', default='gpurun_out/r06_fresh.jsonl')
ap_.add_argument('--save-ap', default=None, help='directory for the per-node APs (.npy per configuration)')
a = ap_.parse_args()

from gem_amd.graph import rmat_graph
from gem_amd.evaluation import reconstruction as gr
from gem_amd.embedding.node2vec import node2vec

gold = {}
for fl in (11, 27):
    p = os.path.join(ROOT, 'tests', 'golden', 'n2v_ref_oracle_rmat%d%s_%s.json' % (a.scale, '' if fl == 11 else '_vocab_order', 'e16k' if a.scale == 17 else 'e128k'))
    if os.path.exists(p):
        gold[fl] = json.load(open(p))
pr = next(iter(gold.values()))['params']
g = rmat_graph(pr['rmat_scale'], pr['edges'], pr['seed'])
nodes = gr.eligible_sample(g, len(next(iter(gold.values()))['ap']))
os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
log = open(a.out, 'a')
cfgs = []
for cfg in a.configs.split(','):            # W:fresh:flags[:hot_count[:neg_count]][*repeats]
    rep = 1
    if '*' in cfg:
        cfg, rep = cfg.split('*'); rep = int(rep)
    cfgs += [cfg] * rep
for cfg in cfgs:
    f = [int(v) for v in cfg.split(':')]
    W, fresh, fl = f[:3]
    hot, negc = (f[3] if len(f) > 3 else -1), (f[4] if len(f) > 4 else 0)
    if fl not in gold:
        continue
    if hot >= 0: os.environ['GEMHIP_SGNS_HOT_COUNT'] = str(hot)
    else: os.environ.pop('GEMHIP_SGNS_HOT_COUNT', None)
    os.environ['GEMHIP_SGNS_NEG_COUNT'] = str(negc)
    ref = gold[fl]
    if W > 0: os.environ['GEMHIP_SGNS_MAX_WAVES'] = str(W)
    else: os.environ.pop('GEMHIP_SGNS_MAX_WAVES', None)
    os.environ['GEMHIP_SGNS_FRESH'] = str(fresh)
    m = node2vec(d=pr['d'], max_iter=1, walk_len=pr['walk_len'], num_walks=pr['num_walks'], con_size=pr['window'], ret_p=1, inout_p=1, seed=20260923, flags=fl)
    t = time.time()
    X = m.learn_embedding(graph=g, is_weighted=True, no_python=True)
    wall = time.time() - t
    apv = gr.sampled_ap_gpu(g, None, X, nodes)
    dd = apv - np.asarray(ref['ap'])
    rec = {'scale': a.scale, 'max_waves': W, 'fresh': fresh, 'flags': fl, 'hot_count': hot, 'neg_count': negc, 'sgns_s': round(m._stats['sgns_seconds'], 3), 'wall_s': round(wall, 2),
           'MAP': float(apv.mean()), 'oracle_MAP': ref['MAP'], 'gap_pct': float(100 * dd.mean() / ref['MAP']),
           'gap_se_pct': float(100 * dd.std(ddof=1) / np.sqrt(len(dd)) / ref['MAP']), 'nodes': int(len(dd)), 'lib': os.environ.get('GEM_HIP_LIB', 'default')}
    s = json.dumps(rec)
    print(s, flush=True); log.write(s + '\n'); log.flush()
    if a.save_ap:
        os.makedirs(a.save_ap, exist_ok=True)
        np.save(os.path.join(a.save_ap, 'ap_scale%d_w%d_fresh%d_f%d.npy' % (a.scale, W, fresh, fl)), apv.astype(np.float32))
