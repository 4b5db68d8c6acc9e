"""Times gemhip_sym_eig_top (the projected eigensolver of the HOPE / LE / LLE solvers: host code, no GPU needed) at 1, 2, 4 and 8
host threads and checks every result against LAPACK.  Record: profiles/r03_host_eig_threads.txt (run on the build container, 8 vCPUs).
    python scripts/bench_host_eig.py [--debug]      (--debug: the library's own phase split on stderr)"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if '--debug' in sys.argv:
    os.environ['GEMHIP_EIG_DEBUG'] = '1'
from gem_amd import _hip  # noqa: E402

L = _hip.lib()


def run(n, m, T, reps=7):
    eff = C.c_int32()
    _hip.check(L.gemhip_set_host_threads(T, C.byref(eff)))
    rs = np.random.RandomState(1)
    U, _ = np.linalg.qr(rs.randn(n, n))
    A0 = (U * np.logspace(0, -6, n)) @ U.T
    A0 = (A0 + A0.T) / 2
    we = np.linalg.eigvalsh(A0)[::-1][:m]
    time.sleep(0.3)                      # let numpy's BLAS worker threads stop spinning: they would occupy the cores the barriers need
    ts = []
    for _ in range(reps):
        A = A0.copy(); w = np.zeros(m); Z = np.zeros((m, n))
        t = time.perf_counter()
        _hip.check(L.gemhip_sym_eig_top(n, _hip.ptr(A, C.c_double), m, _hip.ptr(w, C.c_double), _hip.ptr(Z, C.c_double)))
        ts.append(time.perf_counter() - t)
    print('n %4d  m %3d  threads %d  min %6.2f ms  median %6.2f ms   |w - lapack| %.1e  residual %.1e  orthogonality %.1e' % (
        n, m, eff.value, min(ts) * 1e3, np.median(ts) * 1e3, np.abs(w - we).max(), np.abs(A0 @ Z.T - Z.T * w).max(),
        np.abs(Z @ Z.T - np.eye(m)).max()), flush=True)


if __name__ == '__main__':
    for n, m in ((200, 72), (256, 72), (384, 72), (448, 72), (512, 80)):
        for T in (1, 2, 4, 8):
            run(n, m, T)
