"""Partitioned (episode) schedule on ONE rank at SBM 1M/10M: time of the pair-based pipeline and sampled MAP, next to the
walk-based single-GPU kernel (bench.py) -- the per-GPU cost model of the N-GPU path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from gem_amd import _hip, multi_gpu
from gem_amd.graph import sbm_graph, edge_arrays, to_csr
from gem_amd.evaluation import reconstruction as gr
g = sbm_graph(1000000, 10000000, 100, seed=20260927)
n, src, dst, w, _ = edge_arrays(g); row_ptr, col, ww = to_csr(n, src, dst, w)
b = multi_gpu.HipBackendN2V(n, row_ptr, col, ww, 128)
eps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
job = multi_gpu.Node2VecPartitioned(b, multi_gpu.TorchComm(1), 0, 1, n, 10, 80, 10, 1, seed=1, flags=11, episodes=eps)
torch.cuda.synchronize(); t = time.time(); P = job.run(1.0, 1.0); torch.cuda.synchronize(); el = time.time() - t
nodes = np.random.RandomState(0).choice(n, 256, replace=False)
ap = gr.sampled_ap_gpu(g, None, P.cpu().numpy(), nodes)
print('partitioned world=1 episodes %d: %.2f s for %d pairs (%.2f TB/s algorithmic), sampled MAP %.4f' %
      (eps, el, job.pairs_trained, job.pairs_trained * 7192 / el / 1e12, ap.mean()), flush=True)
