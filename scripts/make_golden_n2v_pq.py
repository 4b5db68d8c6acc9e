#!/usr/bin/env python3
"""second-order (p, q != 1) walks against the SNAP binary.  gem/c_exe/node2vec (argv of gem/embedding/node2vec.py:35-46)
run race-free (OMP_NUM_THREADS=1) on the reference's SBM-1024 test graph with (p, q) = (0.25, 4) and (4, 0.25), d=16, three runs each (the
binary is time-seeded); MAP by the reference-semantics evaluator.  Writes tests/golden/n2v_ref_pq.json."""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from conftest import load_sbm1024
from gem_amd.embedding.node2vec import node2vec
from gem_amd.evaluation import reconstruction as gr
G = load_sbm1024()
n = G.number_of_nodes()
tmp = tempfile.mkdtemp()
gf = os.path.join(tmp, 'g.graph')
with open(gf, 'w') as fh:
    for i, j, w in G.edges(data='weight', default=1):
        fh.write('%d %d %f\n' % (i, j, w))
out = {}
for p, q in ((0.25, 4.0), (4.0, 0.25)):
    maps = []
    for rep in range(3):
        emb = os.path.join(tmp, 'g.emb')
        subprocess.call(['/root/reference/gem/c_exe/node2vec', '-i:' + gf, '-o:' + emb, '-d:16', '-l:80', '-r:10', '-k:10', '-e:1', '-p:%f' % p, '-q:%f' % q,
                         '-dr', '-w'], stdout=subprocess.DEVNULL, env=dict(os.environ, OMP_NUM_THREADS='1'))
        X = np.zeros((n, 16))
        with open(emb) as fh:
            fh.readline()
            for line in fh:
                t = line.split(); X[int(t[0])] = [float(v) for v in t[1:]]
        m = node2vec(d=16, max_iter=1, walk_len=80, num_walks=10, con_size=10, ret_p=p, inout_p=q)
        maps.append(gr.evaluateStaticGraphReconstruction(G, m, X, None)[0])
        print(p, q, maps[-1], flush=True)
    out['sbm1024_d16_t1_p%g_q%g' % (p, q)] = maps
json.dump(out, open(os.path.join(ROOT, 'tests', 'golden', 'n2v_ref_pq.json'), 'w'), indent=1)
