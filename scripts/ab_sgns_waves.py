import sys, time, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from gem_amd import _hip, multi_gpu
from gem_amd.graph import sbm_graph, edge_arrays, to_csr
r = 2
g = sbm_graph(1000000, 10000000, 100, seed=20260927)
n, src, dst, w, _ = edge_arrays(g)
row_ptr, col, ww = to_csr(n, src, dst, w)
b = multi_gpu.HipBackendN2V(n, row_ptr, col, ww, 128)
m = b.num_start_nodes(); b.walks(1.0, 1.0, r, 80, 1, 11, 0, m * r); b.vocab(); b.build_unigram()
for waves in [int(x) for x in sys.argv[1].split(',')]:
    for flags in (43,):
        _hip.check(b.L.gemhip_n2v_set_max_waves(b.h, waves))
        b.init_tables(1); b.pairs(reset=True)
        torch.cuda.synchronize(); t = time.time()
        b.train(10, 1, 0, 0, m * r, m * r * 80, 0, 1, flags)
        torch.cuda.synchronize(); el = time.time() - t
        pairs = b.pairs()
        print('waves', waves, 'flags', flags, 'sgns %.3f s' % el, 'algorithmic TB/s %.3f' % (pairs * 7192 / el / 1e12), flush=True)
