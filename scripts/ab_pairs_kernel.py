"""A/B of the partitioned-schedule pair kernel (sgns_pairs_kernel) with and without the six-way fast path, on ONE rank, at a size where
the tables of a rank live in the Infinity Cache the way an 8-GPU partition of the 1M-node tables does (125k nodes: 64 MB per table)."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from gem_amd import _hip, multi_gpu
from gem_amd.graph import sbm_graph, edge_arrays, to_csr
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
g = sbm_graph(n0, 10 * n0, max(1, n0 // 10000), seed=20260927)
n, src, dst, w, _ = edge_arrays(g); row_ptr, col, ww = to_csr(n, src, dst, w)
for flags, name in ((11, 'fast path'), (11 | 256, 'sequential')):
    b = multi_gpu.HipBackendN2V(n, row_ptr, col, ww, 128)
    job = multi_gpu.Node2VecPartitioned(b, multi_gpu.TorchComm(1), 0, 1, n, 10, 80, 10, 1, seed=1, flags=flags, episodes=16)
    job.run(1.0, 1.0)
    torch.cuda.synchronize(); t = time.time(); job.run(1.0, 1.0); torch.cuda.synchronize(); el = time.time() - t
    ph = job.phase_seconds()
    print('%s: %.3f s for %d pairs, train %.3f s (%.2f G pairs/s)' % (name, el, job.pairs_trained, ph['train'], job.pairs_trained / ph['train'] / 1e9), flush=True)
