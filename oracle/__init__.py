"""oracle -- TEST INFRASTRUCTURE.  CPU restatements of the reference algorithms.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  Nothing under gem_amd/ imports it (tests/test_layout.py checks).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, '_build', 'liboracle.so')
REF_GF = os.path.join(_HERE, '_ref', 'gf')
REF_N2V = os.path.join(_HERE, '_ref', 'node2vec')
_lib = None


def build(force=False):
    """make -C oracle : compiles liboracle.so and (only where /root/reference exists)
    the real reference binaries into oracle/_ref/."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith('.c')]
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(['make', '-s', '-C', _HERE, '_build/liboracle.so'])
    if os.path.exists('/root/reference/gem/c_src/gf.cpp') and not (os.path.exists(REF_GF) and os.path.exists(REF_N2V)):
        subprocess.check_call(['make', '-s', '-C', _HERE, 'ref'])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        i32p, f32p, f64p = C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_double)
        L.oracle_gf_train_f32.argtypes = [C.c_int64, C.c_int64, i32p, i32p, f32p, C.c_int32, C.c_float, C.c_float,
                                          C.c_int32, f32p]
        L.oracle_gf_train_f32.restype = None
        L.oracle_gf_train_f64.argtypes = [C.c_int64, C.c_int64, i32p, i32p, f64p, C.c_int32, C.c_double, C.c_double,
                                          C.c_int32, f64p]
        L.oracle_gf_train_f64.restype = None
        L.oracle_gf_objective.argtypes = [C.c_int64, C.c_int64, i32p, i32p, f32p, C.c_int32, f32p, f64p]
        L.oracle_gf_objective.restype = None
        _lib = L
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def gf_train_f32(n, src, dst, w, d, eta, regu, max_iter, X0):
    """gf.cpp:152-164 in fp32; returns a new array."""
    X = np.ascontiguousarray(X0, dtype=np.float32).copy()
    src = np.ascontiguousarray(src, dtype=np.int32)
    dst = np.ascontiguousarray(dst, dtype=np.int32)
    w = None if w is None else np.ascontiguousarray(w, dtype=np.float32)
    lib().oracle_gf_train_f32(n, len(src), _p(src, C.c_int32), _p(dst, C.c_int32), _p(w, C.c_float), d, eta, regu,
                              max_iter, _p(X, C.c_float))
    return X


def gf_train_f64(n, src, dst, w, d, eta, regu, max_iter, X0):
    """gf.py:93-100 in fp64; returns a new array."""
    X = np.ascontiguousarray(X0, dtype=np.float64).copy()
    src = np.ascontiguousarray(src, dtype=np.int32)
    dst = np.ascontiguousarray(dst, dtype=np.int32)
    w = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
    lib().oracle_gf_train_f64(n, len(src), _p(src, C.c_int32), _p(dst, C.c_int32), _p(w, C.c_double), d, eta, regu,
                              max_iter, _p(X, C.c_double))
    return X


def gf_objective(n, src, dst, w, d, X):
    """gf.cpp:94-113: returns (f1, f2)."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    src = np.ascontiguousarray(src, dtype=np.int32)
    dst = np.ascontiguousarray(dst, dtype=np.int32)
    w = None if w is None else np.ascontiguousarray(w, dtype=np.float32)
    out = np.zeros(2)
    lib().oracle_gf_objective(n, len(src), _p(src, C.c_int32), _p(dst, C.c_int32), _p(w, C.c_float), d, _p(X, C.c_float),
                              _p(out, C.c_double))
    return float(out[0]), float(out[1])
