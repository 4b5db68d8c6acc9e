#!/usr/bin/env python3
"""node2vec reference points at (and near) the BASELINE headline size.

Runs the real SNAP binary (gem/c_exe/node2vec, argv of gem/embedding/node2vec.py:34-53) race-free
(OMP_NUM_THREADS=1) -- or, with --engine oracle, the sequential C restatement oracle/n2v_oracle.c -- on the graph
bench.py builds (gem_amd.graph.sbm_graph, block size 10k, seed 20260923+4), then scores graph-reconstruction MAP
(metrics.computeMAP semantics, gem_amd.evaluation.reconstruction.sampled_map) over a FIXED node sample
(np.random.RandomState(0).choice(n, 1024, replace=False)) and writes tests/golden/n2v_ref_<tag>.json with the MAP,
its standard error and the per-node APs, so that the -m gpu test and bench.py's `quality.reference_map` can compare
the HIP path on the very same graph and sample.

    python scripts/make_golden_n2v_scale.py --nodes 100000 --edges 1000000 --blocks 10           # ~35 min of one core
    python scripts/make_golden_n2v_scale.py --nodes 1000000 --edges 10000000 --blocks 100        # ~5.5 h of one core
"""
import argparse, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gem_amd.graph import sbm_graph, edge_arrays
from gem_amd.evaluation import reconstruction as gr

ap = argparse.ArgumentParser()
ap.add_argument('--nodes', type=int, required=True)
ap.add_argument('--edges', type=int, required=True)
ap.add_argument('--blocks', type=int, required=True)
ap.add_argument('--seed', type=int, default=20260923 + 4)
ap.add_argument('--engine', default='snap', choices=['snap', 'oracle'])
ap.add_argument('--threads', default='1')
ap.add_argument('--p', type=float, default=1.0)
ap.add_argument('--q', type=float, default=1.0)
ap.add_argument('--sample', type=int, default=1024)
ap.add_argument('--tag', default=None)
ap.add_argument('--workdir', default=None)
ap.add_argument('--rmat-scale', type=int, default=0, help='R-MAT graph (gem_amd.graph.rmat_graph(scale, edges, seed)) instead of the SBM')
ap.add_argument('--save-emb', default=None, help='write the trained embedding (float32 .npy) here so a later run can re-score a bigger sample')
ap.add_argument('--load-emb', default=None, help='skip training, score this saved embedding')
ap.add_argument('--eligible-sample', type=int, default=0, help='score this many nodes drawn (RandomState(1), sorted) from the nodes that HAVE a neighbour j > i '
                'instead of --sample uniform nodes: the evaluator only ranks candidates j > i, so on a power-law graph two thirds of a uniform sample score 0 by '
                'construction and the MAP of the rest hangs on a few nodes (gem_amd.evaluation.reconstruction.eligible_sample)')
ap.add_argument('--flags', type=int, default=11, help='oracle engine: 11 = SNAP quirks, unigram table in node-id layout (rounds 2-3); 27 = + the binary\'s table layout (first-appearance order)')
a = ap.parse_args()
PARAMS = dict(n=a.nodes, edges=a.edges, blocks=a.blocks, seed=a.seed, d=128, walk_len=80, num_walks=10, window=10, p=a.p, q=a.q)

if a.rmat_scale:
    from gem_amd.graph import rmat_graph
    g = rmat_graph(a.rmat_scale, a.edges, a.seed)
    PARAMS['rmat_scale'] = a.rmat_scale; PARAMS['n'] = g.n
else:
    g = sbm_graph(a.nodes, a.edges, a.blocks, a.seed)
n = g.n
if a.eligible_sample:
    nodes = gr.eligible_sample(g, a.eligible_sample)
else:
    nodes = np.random.RandomState(0).choice(n, size=min(a.sample, n), replace=False)
tmp = a.workdir or tempfile.mkdtemp(prefix='n2vgold_')
os.makedirs(tmp, exist_ok=True)
print('graph %d nodes %d directed edges, workdir %s' % (n, g.number_of_edges(), tmp), flush=True)

t = time.time()
if a.load_emb:
    X = np.load(a.load_emb)
    prev = json.load(open(a.load_emb + '.json'))
    el, engine = prev['seconds'], prev['engine']
    PARAMS.update(prev['params'])                      # (flags and the graph of the run that made the embedding)
elif a.engine == 'snap':
    gf = os.path.join(tmp, 'g.graph')
    import pandas as pd
    pd.DataFrame({'s': g.src, 'd': g.dst, 'w': np.ones(len(g.src))}).to_csv(gf, sep=' ', header=False, index=False, float_format='%f')
    rc = subprocess.call(['/root/reference/gem/c_exe/node2vec', '-i:' + gf, '-o:' + os.path.join(tmp, 'g.emb'), '-d:%d' % PARAMS['d'],
                          '-l:%d' % PARAMS['walk_len'], '-r:%d' % PARAMS['num_walks'], '-k:%d' % PARAMS['window'], '-e:1',
                          '-p:%f' % a.p, '-q:%f' % a.q, '-dr', '-w'], stdout=subprocess.DEVNULL,
                         env=dict(os.environ, OMP_NUM_THREADS=a.threads))
    el = time.time() - t
    assert rc == 0, rc
    import pandas as pd
    df = pd.read_csv(os.path.join(tmp, 'g.emb'), sep=' ', header=None, skiprows=1, dtype=np.float64)
    df = df.dropna(axis=1, how='all')
    X = np.zeros((n, PARAMS['d']), np.float32)
    X[df[0].to_numpy().astype(np.int64)] = df.iloc[:, 1:1 + PARAMS['d']].to_numpy(dtype=np.float32)
    engine = 'gem/c_exe/node2vec (SNAP ELF), OMP_NUM_THREADS=%s' % a.threads
else:
    import oracle
    _, src, dst, w, _ = edge_arrays(g)
    X = oracle.n2v_train(n, src, dst, None, PARAMS['d'], PARAMS['walk_len'], PARAMS['num_walks'], PARAMS['window'], 1, a.p, a.q, 20260923, a.flags)
    PARAMS['flags'] = a.flags
    if isinstance(X, tuple):
        X = X[0]
    X = np.asarray(X, dtype=np.float32)
    el = time.time() - t
    engine = 'oracle/n2v_oracle.c (sequential restatement of SNAP)' + (', unigram table in the binary\'s vocabulary-order layout' if a.flags & 16 else '')
print('trained in %.0fs' % el, flush=True)
if a.save_emb and not a.load_emb:
    np.save(a.save_emb, X)
    json.dump({'seconds': el, 'engine': engine, 'params': PARAMS}, open(a.save_emb + '.json', 'w'))

sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import score_oracle_ap                                   # batched fp64 scoring, exact tie rule (== reconstruction.average_precision_rows, tests/test_evaluation.py)
order = np.argsort(g.src, kind='stable')
aps = score_oracle_ap.ap_of_nodes(X, g.dst[order].astype(np.int64), np.searchsorted(g.src[order], np.arange(n + 1)), np.asarray(nodes, dtype=np.int64))
aps = np.asarray(aps)
out = {'params': PARAMS, 'engine': engine, 'seconds': el, 'edges_per_s': g.number_of_edges() / el,
       'sample': ('gem_amd.evaluation.reconstruction.eligible_sample(g, %d)' % len(nodes)) if a.eligible_sample else
                 ('np.random.RandomState(0).choice(n, %d, replace=False)' % len(nodes)),
       'MAP': float(aps.mean()), 'MAP_se': float(aps.std(ddof=1) / np.sqrt(len(aps))), 'ap': [round(float(v), 6) for v in aps]}
tag = a.tag or ('%s_%dk' % (a.engine, n // 1000))
path = os.path.join(ROOT, 'tests', 'golden', 'n2v_ref_%s.json' % tag)
json.dump(out, open(path, 'w'))
print(path, 'MAP %.4f +- %.4f, %.0f edges/s' % (out['MAP'], out['MAP_se'], out['edges_per_s']), flush=True)
