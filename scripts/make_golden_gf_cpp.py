"""Golden vectors from the reference's OWN native GF path: oracle/_ref/gf (= g++ -O2 of /root/reference/gem/c_src/gf.cpp) run the way
gem/embedding/gf.py:55-72 runs it -- graph text file (saveGraphToEdgeListTxt's format) in, embedding text file (gf.cpp:115-129) out -- with the clock
its embedding generator is seeded from (gf.cpp:46, std::chrono::system_clock) frozen by oracle/shim/faketime.c (GEM_FAKE_CLOCK), which makes the
binary deterministic.  Writes tests/golden/gf_cpp_binary_<case>.emb (the files the binary wrote, byte for byte) and gf_cpp_binary.json (argv, clock).
Runs in the build container only (needs /root/reference through oracle/_ref/gf); the tests read the committed files."""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle
from conftest import load_karate, load_sbm1024, GOLDEN
from gem_amd.graph import edge_arrays

oracle.build()
assert os.path.exists(oracle.REF_GF) and os.path.exists(oracle.REF_FAKETIME), 'oracle/_ref/gf missing: run in the build container'
CASES = {'karate': dict(graph='karate', d=4, eta=0.05, regu=0.01, max_iter=50, clock='1000.0'),
         'karate_init': dict(graph='karate', d=4, eta=0.05, regu=0.01, max_iter=0, clock='1000.0'),
         'sbm1024': dict(graph='sbm1024', d=16, eta=0.01, regu=0.01, max_iter=20, clock='1700000000.123456789')}
meta = {}
for name, c in CASES.items():
    G = load_karate() if c['graph'] == 'karate' else load_sbm1024()
    n, src, dst, w, _ = edge_arrays(G)
    tmp = tempfile.mkdtemp()
    gfile, efile = os.path.join(tmp, 'g.txt'), os.path.join(GOLDEN, 'gf_cpp_binary_%s.emb' % name)
    with open(gfile, 'w') as fh:                      # graph_util.py:129-134: n, m, then "i j w"
        fh.write('%d\n%d\n' % (n, len(src)))
        for i, j, ww in zip(src.tolist(), dst.tolist(), (w if w is not None else np.ones(len(src))).tolist()):
            fh.write('%d %d %f\n' % (i, j, ww))
    argv = [gfile, efile, '0', '1', str(c['d']), repr(c['eta']), repr(c['regu']), str(c['max_iter']), '10000']
    subprocess.check_call([oracle.REF_GF] + argv, env=dict(os.environ, LD_PRELOAD=oracle.REF_FAKETIME, GEM_FAKE_CLOCK=c['clock']))
    meta[name] = dict(c, n=n, edges=int(len(src)), seed32=oracle.gf_cpp_seed(c['clock']), argv_after_files=argv[2:],
                      weights_all_one=bool(w is None or np.all(np.asarray(w) == 1.0)))
    print(name, 'n', n, 'edges', len(src), 'seed', meta[name]['seed32'], os.path.getsize(efile), 'bytes')
json.dump(meta, open(os.path.join(GOLDEN, 'gf_cpp_binary.json'), 'w'), indent=1)
