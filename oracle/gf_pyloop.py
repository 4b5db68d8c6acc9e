"""oracle/gf_pyloop.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The PYTHON loop of GraphFactorization.learn_embedding (gem/embedding/gf.py:91-101) restated operation by operation in fp64 numpy:
what a GEM user runs when the `gf` executable is missing, and SURVEY 8(d)'s CPU baseline (i) "GEM's CPU learn_embedding()".
The reference file itself cannot travel to the GPU box, so bench.py's `cpu_baseline.python_loop` times THIS (kind "port");
tests/test_oracle_gf.py pins it to the vectors produced by running gf.py itself (tests/golden/gf_*.npz) to 1e-12.

    for _ in range(max_iter):                                  # gf.py:93
        for i, j, w in graph.edges(data='weight', default=1):  # gf.py:94
            if j <= i: continue                                # gf.py:95-96
            term1 = -(w - X[i].X[j]) * X[j]                    # gf.py:97
            term2 = regu * X[i]                                # gf.py:98
            X[i] -= eta * (term1 + term2)                      # gf.py:99-100
"""
import time

import numpy as np


def gf_python_loop(src, dst, w, eta, regu, max_iter, X0, budget_s=None):
    """Returns (X, edge visits done, seconds).  `budget_s`: stop after that many seconds (for timing a bounded sample of a sweep)."""
    X = np.array(X0, dtype=np.float64)
    src = np.asarray(src).tolist(); dst = np.asarray(dst).tolist()
    ww = [1.0] * len(src) if w is None else np.asarray(w, dtype=np.float64).tolist()
    visits = 0
    t0 = time.time()
    for _ in range(max_iter):
        for k in range(len(src)):
            i, j = src[k], dst[k]
            visits += 1
            if j <= i:
                continue
            term1 = -(ww[k] - np.dot(X[i, :], X[j, :])) * X[j, :]
            term2 = regu * X[i, :]
            del_phi = term1 + term2
            X[i, :] -= eta * del_phi
            if budget_s is not None and (visits & 1023) == 0 and time.time() - t0 > budget_s:
                return X, visits, time.time() - t0
    return X, visits, time.time() - t0
