#!/usr/bin/env python3
"""Larger-scale node2vec reference point: the real SNAP binary (gem/c_exe/node2vec, argv of
gem/embedding/node2vec.py:35-46) single-threaded (race-free) on a 16384-node SBM from gem_amd.graph.sbm_graph,
MAP by the vectorised evaluator (pinned equal to the reference evaluator in tests/test_evaluation.py).
Writes tests/golden/n2v_ref_16k.json.  ~5 minutes of CPU."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gem_amd.graph import sbm_graph
from gem_amd.utils import graph_util
from gem_amd.embedding.node2vec import node2vec
from gem_amd.evaluation import reconstruction as gr

PARAMS = dict(n=16384, edges=163840, blocks=8, seed=20260927, d=128, walk_len=80, num_walks=10, window=10)
g = sbm_graph(PARAMS['n'], PARAMS['edges'], PARAMS['blocks'], PARAMS['seed'])
tmp = tempfile.mkdtemp()
gf = os.path.join(tmp, 'g.graph')
with open(gf, 'w') as fh:
    fh.writelines('%d %d %f\n' % (i, j, 1.0) for i, j in zip(g.src.tolist(), g.dst.tolist()))
out = {}
for thr in ('1', '8'):
    t = time.time()
    subprocess.call(['/root/reference/gem/c_exe/node2vec', '-i:' + gf, '-o:' + os.path.join(tmp, 'g.emb'), '-d:%d' % PARAMS['d'],
                     '-l:%d' % PARAMS['walk_len'], '-r:%d' % PARAMS['num_walks'], '-k:%d' % PARAMS['window'], '-e:1', '-p:1.000000',
                     '-q:1.000000', '-dr', '-w'], stdout=subprocess.DEVNULL, env=dict(os.environ, OMP_NUM_THREADS=thr))
    el = time.time() - t
    # isolated nodes never reach the binary: its header counts only the nodes it saw, and GEM's loadEmbedding
    # (graph_util.py:161-169) would index out of bounds; place rows by id into an (n, d) array instead
    X = np.zeros((PARAMS['n'], PARAMS['d']))
    with open(os.path.join(tmp, 'g.emb')) as fh:
        fh.readline()
        for line in fh:
            tok = line.split()
            X[int(tok[0])] = [float(v) for v in tok[1:]]
    m = node2vec(d=PARAMS['d'], max_iter=1, walk_len=80, num_walks=10, con_size=10, ret_p=1, inout_p=1)
    MAP = gr.evaluateStaticGraphReconstruction(g, m, X, None)[0]
    out['t' + thr] = {'MAP': MAP, 'seconds': el, 'edges_per_s': g.number_of_edges() / el}
    print(thr, out['t' + thr], flush=True)
json.dump({'params': PARAMS, 'snap': out}, open(os.path.join(ROOT, 'tests', 'golden', 'n2v_ref_16k.json'), 'w'), indent=1)
