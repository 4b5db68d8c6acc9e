"""Walk matrices of the reference's OWN node2vec binary, made reproducible -> tests/golden/n2v_snap_stream_walks.json.

The prebuilt SNAP ELF (gem/c_exe/node2vec, copied to oracle/_ref/node2vec by oracle/Makefile) seeds its generators with time(NULL).
Under oracle/_ref/libfaketime.so (oracle/shim/faketime.c: time() pinned to $GEM_FAKE_TIME) and OMP_NUM_THREADS=1 it is
deterministic.  It has no option to print its walks, so it runs under the debugger (rocgdb ships with ROCm), stops where
node2vec() hands the finished walk matrix to LearnEmbeddings() (0x40ea30; first argument TVVec<TInt,int64>& = {XDim, YDim, {MxVals,
Vals, ValT*}}) and the matrix is dumped from memory.  Graph files are written the way gem/utils/graph_util.py:137-140
(saveGraphToEdgeListTxtn2v) writes them and the flags are those of gem/embedding/node2vec.py:35-46 (-dr -w).

Needs /root/reference (build container only).  The committed JSON is what tests/test_oracle_n2v.py checks oracle/snap_stream.py
against, bit for bit."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, 'oracle', '_ref')
GDB = '/opt/rocm/bin/rocgdb'


def edge_lines(kind):
    import networkx as nx
    if kind == 'karate':
        g = nx.karate_club_graph().to_directed()
        return ['%d %d %f' % (i, j, 1.0) for i, j in g.edges()]
    if kind == 'karate_weighted':                    # weights 1..4 from a fixed rule; both directions carry the same weight
        g = nx.karate_club_graph().to_directed()
        return ['%d %d %f' % (i, j, 1.0 + ((i * 7 + j * 7 + i * j) % 4)) for i, j in g.edges()]
    if kind == 'directed_with_sinks':                # a directed graph with nodes that have no out-edge (walks end early, zero padding)
        rs = np.random.RandomState(7)
        n = 40
        e = set()
        while len(e) < 120:
            i, j = int(rs.randint(0, n - 6)), int(rs.randint(0, n))
            if i != j:
                e.add((i, j))
        return ['%d %d %f' % (i, j, 1.0) for i, j in sorted(e, key=lambda t: (t[0] * 13 % n, t[1]))]
    raise ValueError(kind)


def run_binary(lines, d, walk_len, num_walks, window, p, q, seed, epochs=1):
    tmp = tempfile.mkdtemp()
    gfile, efile, wfile, cfile = (os.path.join(tmp, f) for f in ('g.graph', 'g.emb', 'walks.bin', 'cmds.gdb'))
    with open(gfile, 'w') as fh:
        fh.write('\n'.join(lines) + '\n')
    argv = ['-i:' + gfile, '-o:' + efile, '-d:%d' % d, '-l:%d' % walk_len, '-r:%d' % num_walks, '-k:%d' % window, '-e:%d' % epochs,
            '-p:%f' % p, '-q:%f' % q, '-dr', '-w']
    with open(cfile, 'w') as fh:
        fh.write('set pagination off\nset environment GEM_FAKE_TIME %d\nset environment LD_PRELOAD %s\nset environment OMP_NUM_THREADS 1\n'
                 'break *0x40ea30\nrun %s\n'
                 'dump binary memory %s *(long*)($rdi+32) (*(long*)($rdi+32))+4*(*(long*)($rdi+24))\n'
                 'printf "DIMS %%ld %%ld\\n", *(long*)$rdi, *(long*)($rdi+8)\ncontinue\nquit\n'
                 % (seed, os.path.join(REF, 'libfaketime.so'), ' '.join(argv), wfile))
    r = subprocess.run([GDB, '-q', '-batch', '-x', cfile, os.path.join(REF, 'node2vec')], capture_output=True, text=True, timeout=600)
    dims = [l for l in r.stdout.splitlines() if l.startswith('DIMS')]
    assert dims, r.stdout[-2000:] + r.stderr[-2000:]
    x, y = (int(v) for v in dims[0].split()[1:])
    walks = np.fromfile(wfile, dtype=np.int32).reshape(x, y)
    emb = open(efile).read() if os.path.exists(efile) else None
    return walks, emb


CASES = [
    # name, graph, p, q, num_walks, walk_len, seed, (d, window, epochs)
    ('karate_p1_q1', 'karate', 1.0, 1.0, 3, 12, 1000, (8, 3, 1)),
    ('karate_p0.25_q4', 'karate', 0.25, 4.0, 3, 12, 1001, (8, 3, 1)),
    ('karate_p4_q0.25', 'karate', 4.0, 0.25, 2, 12, 77, (8, 3, 1)),
    ('karate_weighted_p0.5_q2', 'karate_weighted', 0.5, 2.0, 2, 10, 31337, (8, 3, 1)),
    ('directed_with_sinks_p1_q1', 'directed_with_sinks', 1.0, 1.0, 2, 10, 4242, (8, 3, 1)),
    ('directed_with_sinks_p2_q0.5', 'directed_with_sinks', 2.0, 0.5, 2, 10, 99, (8, 3, 1)),
    # two epochs over 10 200 words each: TrainModel refreshes alpha at words 0, 10 000 and 20 000 against 2 * 10 200 + 1
    ('karate_two_epochs_alpha_schedule', 'karate', 1.0, 1.0, 10, 30, 2024, (4, 2, 2)),
]


def main():
    out = {'_how': 'scripts/make_golden_n2v_snap_stream.py: gem/c_exe/node2vec under oracle/shim/faketime.c (GEM_FAKE_TIME = seed), OMP_NUM_THREADS=1, '
                   'walk matrix dumped with rocgdb at LearnEmbeddings() entry; flags -dr -w as gem/embedding/node2vec.py:35-46', 'cases': {}}
    for name, graph, p, q, r, l, seed, (d, window, epochs) in CASES:
        lines = edge_lines(graph)
        walks, emb = run_binary(lines, d, l, r, window, p, q, seed, epochs)
        again, emb2 = run_binary(lines, d, l, r, window, p, q, seed, epochs)
        assert np.array_equal(walks, again) and emb == emb2, 'the shimmed binary is not deterministic?'
        out['cases'][name] = {'edge_lines': lines, 'p': p, 'q': q, 'num_walks': r, 'walk_len': l, 'seed': seed,
                              'walks': walks.tolist(), 'd': d, 'window': window, 'epochs': epochs, 'emb': emb}
        print(name, walks.shape, 'deterministic: yes')
    path = os.path.join(ROOT, 'tests', 'golden', 'n2v_snap_stream_walks.json')
    with open(path, 'w') as fh:
        json.dump(out, fh)
    print('wrote', path, os.path.getsize(path), 'bytes')


def main_sbm1024():
    """The reference's hyper-parameters (examples/run_sbm.py:70: walk_len 80, num_walks 10, con_size 10, p = q = 1, one epoch) at
    BASELINE's d = 128 on the reference's own SBM-1024 graph (tests/golden/sbm1024_edges.npy = gem/data/sbm.gpickle re-encoded):
    819 200 words.  Kept: the SHA-256 of the binary's walk matrix and its embedding file (float32: the file carries six digits)."""
    import hashlib
    e = np.load(os.path.join(ROOT, 'tests', 'golden', 'sbm1024_edges.npy'))
    lines = ['%d %d %f' % (int(i), int(j), 1.0) for i, j in e.tolist()]
    d, l, r, k, seed = 128, 80, 10, 10, 20260924
    walks, emb = run_binary(lines, d, l, r, k, 1.0, 1.0, seed, 1)
    rows = [ln.split() for ln in emb.strip().split('\n')[1:]]
    ids = np.array([int(x[0]) for x in rows], dtype=np.int32)
    X = np.array([[float(v) for v in x[1:]] for x in rows])
    path = os.path.join(ROOT, 'tests', 'golden', 'n2v_snap_stream_sbm1024.npz')
    np.savez_compressed(path, ids=ids, emb=X.astype(np.float32), walks_sha256=hashlib.sha256(np.ascontiguousarray(walks, dtype=np.int32).tobytes()).hexdigest(),
                        walks_shape=np.array(walks.shape), walks_head=walks[:4], params=json.dumps({'d': d, 'walk_len': l, 'num_walks': r, 'window': k, 'epochs': 1,
                                                                                                     'p': 1.0, 'q': 1.0, 'seed': seed, 'flags': '-dr -w'}))
    print('wrote', path, os.path.getsize(path), 'bytes; walks', walks.shape, 'max |emb|', np.abs(X).max())


if __name__ == '__main__':
    if '--sbm1024' in sys.argv:
        main_sbm1024()
    else:
        main()
