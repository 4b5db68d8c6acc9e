"""HOPE solver knob sweep on SBM 100k/1M d=128: time, restarts, sigma vs the tightest run."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from gem_amd.embedding.hope import HOPE
from gem_amd.graph import sbm_graph
g = sbm_graph(100000, 1000000, 32, seed=20260927)
ref = None
for kw in (dict(tol=1e-7, max_restarts=60), dict(), dict(tol=1e-4), dict(tol=1e-3), dict(krylov_steps=2), dict(krylov_steps=4), dict(oversample=32), dict(oversample=8)):
    for key in ('tol', 'max_restarts', 'krylov_steps', 'oversample'):
        HOPE.hyper_params.pop(key, None)
    m = HOPE(d=128, beta=0.01, **kw)
    t = time.time(); Y = m.learn_embedding(graph=g); el = time.time() - t
    s = m._sigma
    R = None
    if ref is None:
        ref = (s.copy(), Y.copy())
    k = 64
    # subspace agreement of U with the reference run: ||U_ref^T U||_F^2 / k
    U = Y[:, :k] / np.sqrt(s); Ur = ref[1][:, :k] / np.sqrt(ref[0])
    ov = np.linalg.norm(Ur.T @ U) ** 2 / k
    ov32 = np.linalg.norm(Ur[:, 32:].T @ U[:, 32:]) ** 2 / 32
    print(kw, 'wall %.3f dev %.3f eig %.3f (%d) restarts %d basis %d spmm %d | sigma rel err max %.2e | subspace overlap all %.5f top32 %.6f'
          % (el, m._stats['device_seconds'], m._stats['host_eig_seconds'], m._stats['host_eig_calls'], m._stats['restarts'], m._stats['basis_columns'],
             m._stats['spmm_launches'], np.abs(s / ref[0] - 1).max(), ov, ov32), flush=True)
