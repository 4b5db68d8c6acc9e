#!/usr/bin/env python3
"""The sequential node2vec oracle (oracle/n2v_oracle.c) on an R-MAT graph in RESUMABLE chunks: hour-long runs do not survive a restart of the build
session, so the pass is cut into chunks of walks -- oracle.sgns_train[_vocab_order] over walks [a, z) with token_offset = a * walk_len and
walk_id_offset = a is exactly the corresponding stretch of the one-call pass (same Philox keys per walk id, same alpha per token index; --selftest
checks it bit for bit on a small graph) -- and both tables are checkpointed after every chunk.  Re-running the same command resumes.

    python scripts/oracle_n2v_resumable.py --rmat-scale 20 --edges 16000000 --flags 27 --out .refruns/oracle_rmat20_f27.npy
writes <out> (float32 embedding) and <out>.json like scripts/make_golden_n2v_scale.py --save-emb; score it with --load-emb there."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
from gem_amd.graph import rmat_graph, edge_arrays

ap = argparse.ArgumentParser()
ap.add_argument('--rmat-scale', type=int, default=20)
ap.add_argument('--edges', type=int, default=16000000)
ap.add_argument('--seed', type=int, default=20260928)
ap.add_argument('--flags', type=int, default=27)
ap.add_argument('--out', default=None)
ap.add_argument('--chunk-walks', type=int, default=150000)
ap.add_argument('--selftest', action='store_true')
ap.add_argument('--sbm', default=None, help='nodes,edges,blocks: an SBM (gem_amd.graph.sbm_graph with --seed) instead of the R-MAT graph')
ap.add_argument('--train-seed', type=int, default=20260923, help='the Philox seed of walks, initial table and draws (the seed the GPU tests train with)')
ap.add_argument('--wide', action='store_true', help='oracle_sgns_train_wide: the dot product in 32 partial sums (about 4x the speed; the engine string says so)')
ap.add_argument('--walks-cache', default=None, help='.npy the walk matrix is written to / memory-mapped from (the two layouts train on the same walks: one copy in the page cache)')
a = ap.parse_args()
D, L, R, WIN, SEED = 128, 80, 10, 10, a.train_seed


def run(n, src, dst, flags, out, chunk, d=D, l=L, r=R, wide=False, walks_cache=None):
    if walks_cache and os.path.exists(walks_cache):
        walks = np.load(walks_cache, mmap_mode='r')
    else:
        rp, cs, _ = oracle.sorted_csr(n, src, dst, None)
        walks = oracle.n2v_walks(rp, cs, None, None, 1.0, 1.0, r, l, SEED, flags & ~16)     # (bit 16 is the unigram layout: the walks do not see it)
        if walks_cache:
            np.save(walks_cache + '.tmp.npy', walks); os.replace(walks_cache + '.tmp.npy', walks_cache)
            walks = np.load(walks_cache, mmap_mode='r')
    tot = walks.size
    cnt = oracle.n2v_vocab(n, walks)
    if flags & 16:
        slot_tab, UTn, KTn = oracle.unigram_build_vocab_order(cnt, walks, flags)[:3]
    else:
        UT, KT = oracle.unigram_build(cnt)
    ck = out + '.ckpt'
    if os.path.exists(ck + '.json'):
        st = json.load(open(ck + '.json'))
        P = np.load(ck + '.P.npy'); N = np.load(ck + '.N.npy')
        done, secs = st['walks_done'], st['seconds']
        print('resuming at walk %d of %d (%.0f s so far)' % (done, len(walks), secs), flush=True)
    else:
        P, N = oracle.sgns_init(n, d, SEED)
        done, secs = 0, 0.0
    while done < len(walks):
        z = min(len(walks), done + chunk)
        t = time.time()
        w = np.ascontiguousarray(walks[done:z])
        if wide:
            if flags & 16: oracle.sgns_train_wide(w, WIN, 0.025, 1, 0, tot, done * l, done, slot_tab, UTn, KTn, SEED, flags, P, N)
            else: oracle.sgns_train_wide(w, WIN, 0.025, 1, 0, tot, done * l, done, None, UT, KT, SEED, flags, P, N)
        elif flags & 16:
            oracle.sgns_train_vocab_order(w, WIN, 0.025, 1, 0, tot, done * l, done, slot_tab, UTn, KTn, SEED, flags, P, N)
        else:
            oracle.sgns_train(w, WIN, 0.025, 1, 0, tot, done * l, done, UT, KT, SEED, flags, P, N)
        secs += time.time() - t
        done = z
        np.save(ck + '.P.tmp.npy', P); np.save(ck + '.N.tmp.npy', N)
        os.replace(ck + '.P.tmp.npy', ck + '.P.npy'); os.replace(ck + '.N.tmp.npy', ck + '.N.npy')
        json.dump({'walks_done': done, 'seconds': secs}, open(ck + '.json.tmp', 'w')); os.replace(ck + '.json.tmp', ck + '.json')
        print('walk %d of %d, %.0f s' % (done, len(walks), secs), flush=True)
    return P, secs


if a.selftest:
    g = rmat_graph(10, 12000, 5)
    n, src, dst, _, _ = edge_arrays(g)
    import tempfile
    for flags in (11, 27):
        tmp = os.path.join(tempfile.mkdtemp(), 'x.npy')
        P, _ = run(n, src, dst, flags, tmp, 777, d=16, l=30, r=2)
        X, _ = oracle.n2v_train(n, src, dst, None, 16, 30, 2, WIN, 1, 1.0, 1.0, SEED, flags)
        assert np.array_equal(P, X), flags
        Pw, _ = run(n, src, dst, flags, tmp + '.w.npy', 777, d=32, l=30, r=2, wide=True, walks_cache=tmp + '.walks.npy')
        Pw1, _ = run(n, src, dst, flags, tmp + '.w1.npy', 10 ** 9, d=32, l=30, r=2, wide=True, walks_cache=tmp + '.walks.npy')
        assert np.array_equal(Pw, Pw1), flags
    print('selftest ok: chunked == one call, bit for bit, both layouts')
    sys.exit(0)

if a.sbm:
    from gem_amd.graph import sbm_graph
    nn, ee, bb = [int(v) for v in a.sbm.split(',')]
    g = sbm_graph(nn, ee, bb, a.seed)
else:
    g = rmat_graph(a.rmat_scale, a.edges, a.seed)
n, src, dst, _, _ = edge_arrays(g)
P, secs = run(n, src, dst, a.flags, a.out, a.chunk_walks, wide=a.wide, walks_cache=a.walks_cache)
np.save(a.out, np.asarray(P, dtype=np.float32))
PARAMS = dict(n=g.n, edges=a.edges, blocks=1, seed=a.seed, d=D, walk_len=L, num_walks=R, window=WIN, p=1.0, q=1.0, rmat_scale=a.rmat_scale, flags=a.flags, train_seed=SEED)
if a.sbm:
    PARAMS.update(edges=ee, blocks=bb); del PARAMS['rmat_scale']
engine = 'oracle/n2v_oracle.c (sequential restatement of SNAP)' + (', unigram table in the binary\'s vocabulary-order layout' if a.flags & 16 else '') + \
         (', oracle_sgns_train_wide (dot product in 32 interleaved partial sums)' if a.wide else '') + \
         '; run in resumable chunks (scripts/oracle_n2v_resumable.py)'
json.dump({'seconds': secs, 'engine': engine, 'params': PARAMS}, open(a.out + '.json', 'w'))
for f in ('.ckpt.P.npy', '.ckpt.N.npy', '.ckpt.json'):
    os.remove(a.out + f)
print('done in %.0f s of oracle time' % secs, flush=True)
