#!/usr/bin/env python3
"""Score a SAVED sequential-oracle embedding of an R-MAT graph with the GPU evaluator (gem_amd/csrc/eval.hip: metrics.computeMAP semantics per sampled node, fp64
dot products of the fp32 rows, ties by node id -- node by node equal to the CPU scorer scripts/score_oracle_ap.py, tests/test_eval_gpu.py) and write the golden
tests/test_rmat_gpu.py and bench.py pair Hogwild launches with.  For scale 22 the CPU scorer would need ~13 core-hours per layout (a 4.2M-element sort per
node); the embedding (2.1 GB) travels to the GPU box once instead.  Optionally also runs Hogwild launches at given widths and prints the paired gaps.

    python scripts/gpu_score_oracle.py --emb oracle_push/oracle_rmat22_f27.npy --scale 22 --flags 27 --out gpurun_out/r06l --widths 0,768
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gem_amd.graph import rmat_graph
from gem_amd.evaluation import reconstruction as gr

ap_ = argparse.ArgumentParser()
ap_.add_argument('--emb', required=True)
ap_.add_argument('--scale', type=int, required=True)
ap_.add_argument('--edges', type=int, default=0)
ap_.add_argument('--flags', type=int, required=True)
ap_.add_argument('--sample', type=int, default=131072)
ap_.add_argument('--out', default='gpurun_out/r06l')
ap_.add_argument('--widths', default='', help='comma-separated GEMHIP_SGNS_MAX_WAVES values (0 = the planner) for paired Hogwild launches')
a = ap_.parse_args()
edges = a.edges or {22: 64000000, 20: 16000000, 17: 2000000}[a.scale]
os.makedirs(a.out, exist_ok=True)
side = json.load(open(a.emb + '.json'))
g = rmat_graph(a.scale, edges, 20260928)
nodes = gr.eligible_sample(g, a.sample)
X = np.load(a.emb)
t = time.time()
aps = gr.sampled_ap_gpu(g, None, X, nodes)
print('scored %d nodes in %.1f s' % (len(nodes), time.time() - t), flush=True)
del X
name = 'n2v_ref_oracle_rmat%d%s_e%dk.json' % (a.scale, '_vocab_order' if a.flags & 16 else '', len(nodes) // 1024)
out = {'params': side['params'], 'engine': side['engine'], 'seconds': side['seconds'], 'edges_per_s': g.number_of_edges() / side['seconds'],
       'sample': 'gem_amd.evaluation.reconstruction.eligible_sample(g, %d)  [scored on the GPU box from the saved embedding by scripts/gpu_score_oracle.py: '
                 'sampled_ap_gpu, the evaluator the launches are scored with]' % len(nodes),
       'MAP': float(aps.mean()), 'MAP_se': float(aps.std(ddof=1) / np.sqrt(len(aps))), 'ap': [round(float(v), 6) for v in aps]}
assert out['params']['flags'] == a.flags and out['params'].get('rmat_scale') == a.scale
json.dump(out, open(os.path.join(a.out, name), 'w'))
print(name, 'MAP %.6f +- %.6f' % (out['MAP'], out['MAP_se']), flush=True)
if a.widths:
    from gem_amd.embedding.node2vec import node2vec
    pr = out['params']
    log = open(os.path.join(a.out, 'paired_rmat%d_f%d.jsonl' % (a.scale, a.flags)), 'a')
    for W in [int(v) for v in a.widths.split(',')]:
        if W > 0: os.environ['GEMHIP_SGNS_MAX_WAVES'] = str(W)
        else: os.environ.pop('GEMHIP_SGNS_MAX_WAVES', None)
        m = node2vec(d=pr['d'], max_iter=1, walk_len=pr['walk_len'], num_walks=pr['num_walks'], con_size=pr['window'], ret_p=1, inout_p=1, seed=pr.get('train_seed', 20260923), flags=a.flags)
        Y = m.learn_embedding(graph=g, is_weighted=True, no_python=True)
        apv = gr.sampled_ap_gpu(g, None, Y, nodes)
        dd = apv - aps
        rec = {'scale': a.scale, 'flags': a.flags, 'max_waves': W, 'sgns_s': round(m._stats['sgns_seconds'], 2), 'MAP': float(apv.mean()), 'oracle_MAP': out['MAP'],
               'gap_pct': float(100 * dd.mean() / aps.mean()), 'gap_se_pct': float(100 * dd.std(ddof=1) / np.sqrt(len(dd)) / aps.mean())}
        s = json.dumps(rec); print(s, flush=True); log.write(s + '\n'); log.flush()
