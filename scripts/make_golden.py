#!/usr/bin/env python3
"""Generate tests/golden/* by RUNNING THE REFERENCE in this container.

Run once (here, where /root/reference exists):  python scripts/make_golden.py
The outputs are small data fixtures committed under tests/golden/; the GPU box
never needs /root/reference.

What is produced, and from which reference code:
  karate.edgelist, sbm1024_edges.npy, sbm1024_labels.npy
        <- tests/data/{karate.edgelist,sbm.gpickle,sbm_node_labels.pickle}
           (data fixtures; the gpickle is a networkx-1.x pickle, re-encoded as
           an int32 edge array in *reference iteration order*)
  ref_karate_{HOPE,GraphFactorization,node2vec}.txt, ref_sbm_GraphFactorization.npz
        <- tests/karate_res/*.txt, tests/smb_res/GraphFactorization.txt
           (the reference's own golden vectors)
  hope_*.npz   <- gem/embedding/hope.py:23-41 executed literally
                  (nx.to_numpy_matrix shimmed: removed in networkx 3)
  gf_*.npz     <- gem/embedding/gf.py:91-101 executed literally under np.random.seed
  n2v_ref.json <- gem/c_exe/node2vec (SNAP ELF) driven exactly like
                  gem/embedding/node2vec.py:34-48, MAP by the reference evaluator
  map_ref.json <- gem/evaluation/evaluate_graph_reconstruction.py:8-46 on the goldens
                  (pins gem_amd.evaluation's vectorised MAP)
"""
import io
import json
import os
import pickle
import shutil
import subprocess
import sys
import tempfile
import contextlib

REF = '/root/reference'
sys.path.insert(0, REF)
os.environ.setdefault('MPLBACKEND', 'Agg')

import numpy as np
import networkx as nx

if not hasattr(nx, 'to_numpy_matrix'):          # removed in networkx 3
    nx.to_numpy_matrix = lambda g, *a, **k: np.asmatrix(nx.to_numpy_array(g, *a, **k))

from gem.utils import graph_util                                      # noqa: E402
from gem.embedding.hope import HOPE                                    # noqa: E402
from gem.embedding.gf import GraphFactorization                        # noqa: E402
from gem.embedding.node2vec import node2vec                            # noqa: E402
from gem.evaluation import evaluate_graph_reconstruction as gr         # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')
OUT = os.path.abspath(OUT)
os.makedirs(OUT, exist_ok=True)


def load_karate():
    G = graph_util.loadGraphFromEdgeListTxt(os.path.join(REF, 'tests/data/karate.edgelist'), directed=True)
    return G.to_directed()


def load_sbm():
    with open(os.path.join(REF, 'tests/data/sbm.gpickle'), 'rb') as f:
        G = pickle.load(f)
    H = nx.DiGraph()                       # exactly tests/test_sbm.py:33-40
    H.add_nodes_from(G.__dict__['node'])
    for s in G.__dict__['edge'].keys():
        for t in G.__dict__['edge'][s].keys():
            H.add_edge(s, t)
    return H


def edges_array(G):
    return np.array([(i, j) for i, j in G.edges()], dtype=np.int32)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def ref_map(G, model, X):
    MAP, prec, _, _ = gr.evaluateStaticGraphReconstruction(G, model, X, None)
    return float(MAP), [float(p) for p in prec[:20]]


def main():
    kar = load_karate()
    sbm = load_sbm()
    shutil.copy(os.path.join(REF, 'tests/data/karate.edgelist'), os.path.join(OUT, 'karate.edgelist'))
    np.save(os.path.join(OUT, 'sbm1024_edges.npy'), edges_array(sbm))
    np.save(os.path.join(OUT, 'sbm1024_nodes.npy'), np.array(list(sbm.nodes()), dtype=np.int32))
    with open(os.path.join(REF, 'tests/data/sbm_node_labels.pickle'), 'rb') as f:
        lab = pickle.load(f, encoding='latin1')
    np.save(os.path.join(OUT, 'sbm1024_labels.npy'), np.asarray(lab.toarray().argmax(1)).ravel().astype(np.int16))
    for name in ('HOPE', 'GraphFactorization', 'node2vec', 'LaplacianEigenmaps', 'LocallyLinearEmbedding'):
        shutil.copy(os.path.join(REF, 'tests/karate_res/%s.txt' % name), os.path.join(OUT, 'ref_karate_%s.txt' % name))
    tgt = np.loadtxt(os.path.join(REF, 'tests/smb_res/GraphFactorization.txt'))
    np.savez_compressed(os.path.join(OUT, 'ref_sbm_GraphFactorization.npz'), X=tgt.astype(np.float32))

    maps = {}
    # ---------------- HOPE: literal hope.py ----------------
    m = HOPE(d=4, beta=0.01)
    Xk = quiet(m.learn_embedding, graph=kar, is_weighted=True, no_python=True)
    maps['karate_hope_fresh'] = ref_map(kar, m, Xk)[0]
    m_g = HOPE(d=4, beta=0.01)
    maps['karate_hope_golden'] = ref_map(kar, m_g, np.loadtxt(os.path.join(OUT, 'ref_karate_HOPE.txt')))[0]
    np.savez_compressed(os.path.join(OUT, 'hope_karate_d4.npz'), X=np.asarray(Xk), beta=0.01,
                        nodes=np.array(list(kar.nodes()), dtype=np.int32))
    for d in (32,):
        m = HOPE(d=d, beta=0.01)
        Xs = quiet(m.learn_embedding, graph=sbm, is_weighted=True, no_python=True)
        maps['sbm1024_hope_d%d' % d] = ref_map(sbm, m, Xs)[0]
        np.savez_compressed(os.path.join(OUT, 'hope_sbm1024_d%d.npz' % d), X=np.asarray(Xs), beta=0.01)
    # singular values of S for sbm1024 (for sigma parity at other k)
    A = np.asarray(nx.to_numpy_array(sbm))
    S = np.linalg.inv(np.eye(A.shape[0]) - 0.01 * A) @ (0.01 * A)
    sv = np.linalg.svd(S, compute_uv=False)
    np.save(os.path.join(OUT, 'hope_sbm1024_sigma.npy'), sv[:160])

    # ---------------- GF: literal gf.py loop under a fixed numpy seed ----------------
    def gf_run(G, tag, seed, **hp):
        np.random.seed(seed)
        m = GraphFactorization(data_set='golden', **hp)
        X = quiet(m.learn_embedding, graph=G, is_weighted=True, no_python=False)
        np.random.seed(seed)
        X0 = 0.01 * np.random.randn(len(G.nodes), hp['d'])
        np.savez_compressed(os.path.join(OUT, 'gf_%s.npz' % tag), X=X, X0=X0, seed=seed,
                            **{k: v for k, v in hp.items()})
        return m, X
    m, X = gf_run(kar, 'karate_ref_hp', 11, d=2, max_iter=300, eta=1e-4, regu=1.0)
    maps['karate_gf_ref_hp'] = ref_map(kar, m, X)[0]
    m, X = gf_run(kar, 'karate_train', 12, d=8, max_iter=400, eta=0.05, regu=0.01)
    maps['karate_gf_train'] = ref_map(kar, m, X)[0]
    m, X = gf_run(sbm, 'sbm1024_d32', 13, d=32, max_iter=5, eta=0.02, regu=0.01)
    maps['sbm1024_gf_d32_5sweeps'] = ref_map(sbm, m, X)[0]
    mg = GraphFactorization(d=2, max_iter=1, eta=1e-4, regu=1.0, data_set='golden')
    maps['karate_gf_golden'] = ref_map(kar, mg, np.loadtxt(os.path.join(OUT, 'ref_karate_GraphFactorization.txt')))[0]
    mg = GraphFactorization(d=128, max_iter=1, eta=1e-4, regu=1.0, data_set='golden')
    maps['sbm1024_gf_golden'] = ref_map(sbm, mg, tgt)[0]

    # ---------------- node2vec: the SNAP ELF, argv of node2vec.py:35-46 ----------------
    # The binary shares ONE time-seeded TRnd between its OpenMP threads without locking; with
    # several threads the racing generator visibly degrades the embedding (SBM-1024 MAP 0.10
    # vs 0.18 single-threaded).  Both are recorded: *_t1 = OMP_NUM_THREADS=1 (the race-free
    # meaning of the algorithm, the parity target), *_t8 = 8 threads as GEM runs it by default.
    n2v = {}
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)
    try:
        import time
        for thr in ('1', '8'):
            os.environ['OMP_NUM_THREADS'] = thr
            for G, gname, d, reps in ((kar, 'karate', 2, 5), (sbm, 'sbm1024', 16, 3), (sbm, 'sbm1024', 128, 2)):
                key = '%s_d%d_t%s' % (gname, d, thr)
                n2v[key] = []
                for r in range(reps):
                    m = node2vec(d=d, max_iter=1, walk_len=80, num_walks=10, con_size=10, ret_p=1, inout_p=1)
                    X = quiet(m.learn_embedding, graph=G, is_weighted=True, no_python=True)
                    n2v[key].append(ref_map(G, m, X)[0])
                    if r == 0 and gname == 'sbm1024':
                        np.savez_compressed(os.path.join(OUT, 'n2v_snap_%s.npz' % key), X=X.astype(np.float32))
                    time.sleep(1.1)                   # the ELF seeds with time()
    finally:
        os.environ.pop('OMP_NUM_THREADS', None)
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)
    mg = node2vec(d=2, max_iter=1, walk_len=80, num_walks=10, con_size=10, ret_p=1, inout_p=1)
    maps['karate_n2v_golden'] = ref_map(kar, mg, np.loadtxt(os.path.join(OUT, 'ref_karate_node2vec.txt')))[0]
    with open(os.path.join(OUT, 'n2v_ref.json'), 'w') as f:
        json.dump(n2v, f, indent=1)
    with open(os.path.join(OUT, 'map_ref.json'), 'w') as f:
        json.dump(maps, f, indent=1)
    print(json.dumps(maps, indent=1))
    print(json.dumps(n2v, indent=1))


if __name__ == '__main__':
    main()
