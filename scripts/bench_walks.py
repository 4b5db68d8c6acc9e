"""Walk-kernel throughput on SBM 1M/10M: p=q=1 vs second-order (rejection sampling), unit vs random weights."""
import sys, time, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from gem_amd import _hip, multi_gpu
from gem_amd.graph import sbm_graph, edge_arrays, to_csr
g = sbm_graph(1000000, 10000000, 100, seed=20260927)
n, src, dst, w, _ = edge_arrays(g)
for weighted in (False, True):
    ww = (np.random.RandomState(1).rand(len(src)) + 0.5).astype(np.float32) if weighted else None
    row_ptr, col, wv = to_csr(n, src, dst, ww)
    b = multi_gpu.HipBackendN2V(n, row_ptr, col, wv, 8)
    t = time.time(); _hip.check(b.L.gemhip_n2v_build_alias(b.h, None)); torch.cuda.synchronize(); ta = time.time() - t
    for p, q in ((1.0, 1.0), (0.25, 4.0), (4.0, 0.25)):
        for rep in range(2):
            torch.cuda.synchronize(); t = time.time()
            b.walks(p, q, 10, 80, 1, 11, 0, b.num_start_nodes() * 10)
            torch.cuda.synchronize(); el = time.time() - t
        print('weighted', weighted, 'alias build %.1f ms' % (ta * 1e3), 'p', p, 'q', q, 'walks %.1f ms' % (el * 1e3), '%.1f G steps/s' % (n * 10 * 79 / el / 1e9), flush=True)
    b.close()
