"""CPU tests: libgem_hip.so loads and exports exactly what include/gem_hip.h declares.
No compute calls here (no GPU in the CPU tier)."""
import ctypes
import os
import re

import pytest

from gem_amd import _hip, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'gem_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(gemhip_[a-z0-9_]+)\s*\(', txt)))


def test_library_builds_and_exports_every_declared_symbol():
    build.build()
    L = ctypes.CDLL(_hip.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(L, s), 'include/gem_hip.h declares %s but libgem_hip.so does not export it' % s


def test_ctypes_binding_covers_the_header():
    assert _hip.declared_symbols() == header_symbols()
    assert _hip.lib().gemhip_version() == 100


def header_prototypes():
    """name -> list of C parameter type strings, parsed from include/gem_hip.h."""
    txt = open(os.path.join(ROOT, 'include', 'gem_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    out = {}
    for m in re.finditer(r'\b(?:int|const char \*)\s*(gemhip_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;', txt, flags=re.S):
        params = ' '.join(m.group(2).split())
        if re.match(r'int \(\*\w+\)\(', params):                 # one parameter that is a function pointer
            out[m.group(1)] = ['fnptr']
            continue
        plist = [] if params in ('', 'void') else [q.strip() for q in params.split(',')]
        out[m.group(1)] = plist
    return out


def test_ctypes_signatures_match_the_header_prototypes():
    """Arity and the pointer / integer / float class of every parameter: an ABI drift between include/gem_hip.h and the
    ctypes table would corrupt arguments silently."""
    protos = header_prototypes()
    assert sorted(protos) == header_symbols()

    def klass_c(t):
        if t == 'fnptr' or '*' in t or re.search(r'\bgemhip_\w+_t\b', t):
            return 'ptr'
        if re.match(r'(const )?(float)\b', t):
            return 'f32'
        if re.match(r'(const )?(double)\b', t):
            return 'f64'
        m = re.match(r'(const )?(u?int(32|64)_t|int)\b', t)
        assert m, t
        return {'int': 'i32', 'int32_t': 'i32', 'uint32_t': 'i32', 'int64_t': 'i64', 'uint64_t': 'i64'}[m.group(2)]

    def klass_py(a):
        if a in (ctypes.c_float,):
            return 'f32'
        if a in (ctypes.c_double,):
            return 'f64'
        if a in (ctypes.c_int, ctypes.c_int32, ctypes.c_uint32):
            return 'i32'
        if a in (ctypes.c_int64, ctypes.c_uint64):
            return 'i64'
        return 'ptr'

    for name, (res, args) in _hip._SIGS.items():
        want = [klass_c(t) for t in protos[name]]
        got = [klass_py(a) for a in args]
        assert got == want, '%s: header %s vs ctypes %s' % (name, want, got)


def test_no_silent_fallback_without_gpu():
    """On a machine without a HIP device the product path must fail loudly."""
    import numpy as np
    from gem_amd.embedding.gf import GraphFactorization
    from gem_amd.graph import EdgeListGraph
    if _hip.device_count() > 0:
        pytest.skip('GPU present')
    g = EdgeListGraph(3, [0, 1], [1, 2])
    with pytest.raises(_hip.GemHipError):
        GraphFactorization(d=2, eta=0.1, regu=0.1, max_iter=1).learn_embedding(graph=g)


def test_product_never_imports_oracle():
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, 'gem_amd')):
        for f in fs:
            if f.endswith(('.py', '.hip', '.hpp', '.cpp', '.h')):
                t = open(os.path.join(dp, f)).read()
                if re.search(r'^\s*(import|from)\s+oracle\b', t, flags=re.M) or 'liboracle' in t or '_ref/' in t:
                    bad.append(f)
    assert not bad, bad


def test_host_symmetric_eigensolver():
    """gemhip_sym_eig is host code (tred2/tql2) -- the projected eigenproblems of HOPE run through it."""
    import numpy as np
    L = _hip.lib()
    _hip._register_lapack(L)                       # exercise the optional host-eigensolver hook too
    rng = np.random.RandomState(0)
    for n in (1, 2, 7, 34, 96, 128, 257):
        M = rng.randn(n, n)
        A = (M + M.T) / 2 if n != 34 else M[:, :10] @ M[:, :10].T          # n=34: rank 10
        V = A.copy(); w = np.zeros(n)
        fn = L.gemhip_sym_eig_builtin if n % 2 else L.gemhip_sym_eig      # built-in QL and the LAPACK callback route
        _hip.check(fn(n, _hip.ptr(V, ctypes.c_double), _hip.ptr(w, ctypes.c_double)))
        scale = max(np.abs(A).max(), 1.0)
        assert np.all(np.diff(w) >= 0)
        assert np.abs(w - np.linalg.eigvalsh(A)).max() < 1e-11 * scale * n
        assert np.abs(V @ np.diag(w) @ V.T - A).max() < 1e-11 * scale * n
        assert np.abs(V.T @ V - np.eye(n)).max() < 1e-12 * n
    L.gemhip_set_sym_eig_callback(None)


def test_host_partial_eigensolver():
    """gemhip_sym_eig_top (Householder + QL eigenvalues + inverse iteration + back-transformation of m vectors): the
    Rayleigh-Ritz step of the Krylov solver.  Checked against LAPACK on separated, clustered, repeated and rank-deficient
    spectra -- the cases inverse iteration is known to need care with."""
    import numpy as np
    L = _hip.lib()
    rng = np.random.RandomState(0)

    def check(A, m):
        n = A.shape[0]
        V = np.ascontiguousarray(A, dtype=np.float64).copy(); w = np.zeros(m); Z = np.zeros((m, n))
        _hip.check(L.gemhip_sym_eig_top(n, _hip.ptr(V, ctypes.c_double), m, _hip.ptr(w, ctypes.c_double), _hip.ptr(Z, ctypes.c_double)))
        Z = Z.T
        wr = np.linalg.eigvalsh(A)[::-1][:m]
        scale = max(np.abs(wr).max(), 1e-300)
        assert np.abs(w - wr).max() <= 1e-12 * scale
        assert np.abs(A @ Z - Z * w).max() <= 1e-10 * scale
        assert np.abs(Z.T @ Z - np.eye(m)).max() <= 1e-9

    for n, m in ((290, 80), (200, 48), (128, 30), (5, 2), (3, 3), (2, 1), (1, 1)):
        B = rng.randn(3 * n + 5, n)
        check(B.T @ B / n, m)
    run_cases(check, rng)
    # the same cases with the reduction and the back-transformation on ONE thread and on FOUR (gemhip_set_host_threads): from n = 192 on
    # the Householder steps are shared between threads (block-cyclic columns, two barriers per step)
    eff = ctypes.c_int32()
    for T in (1, 4):
        _hip.check(L.gemhip_set_host_threads(T, ctypes.byref(eff)))
        assert eff.value == T
        run_cases(check, np.random.RandomState(1))
        for n, m in ((192, 60), (193, 20), (448, 72), (512, 80)):
            B = rng.randn(2 * n, n)
            check(B.T @ B / n, m)
    _hip.check(L.gemhip_set_host_threads(0, ctypes.byref(eff)))      # back to the default
    assert 1 <= eff.value <= 4
    assert L.gemhip_set_host_threads(17, None) != 0
    assert L.gemhip_sym_eig_top(4, None, 2, None, None) != 0


def test_threaded_reduction_does_not_depend_on_when_its_threads_are_called_off(monkeypatch):
    """eig_reduce_mt times its steps and calls the threads off on a contended host; the remaining steps run the SAME arithmetic on one
    thread (eig_reduce_virtual), so the result is a function of the matrix and the thread count only -- bit for bit -- wherever the switch
    happens (GEMHIP_EIG_TEST_BAIL_AFTER forces it after k steps)."""
    import numpy as np
    L = _hip.lib()
    rng = np.random.RandomState(3)

    def solve(A, m):
        n = A.shape[0]
        V = A.copy(); w = np.zeros(m); Z = np.zeros((m, n))
        _hip.check(L.gemhip_sym_eig_top(n, _hip.ptr(V, ctypes.c_double), m, _hip.ptr(w, ctypes.c_double), _hip.ptr(Z, ctypes.c_double)))
        return w, Z
    try:
        for n in (192, 300):
            B = rng.randn(2 * n, n)
            half = np.zeros((n, n)); half[:n // 2, :n // 2] = (B.T @ B)[:n // 2, :n // 2]
            for A in (B.T @ B / n, np.diag(rng.rand(n)), half):
                for T in (2, 4):
                    _hip.check(L.gemhip_set_host_threads(T, None))
                    monkeypatch.delenv('GEMHIP_EIG_TEST_BAIL_AFTER', raising=False)
                    w0, Z0 = solve(A, 40)
                    for k in (0, 1, 7, 50):
                        monkeypatch.setenv('GEMHIP_EIG_TEST_BAIL_AFTER', str(k))
                        w, Z = solve(A, 40)
                        assert np.array_equal(w, w0) and np.array_equal(Z, Z0), (n, T, k)
    finally:
        monkeypatch.delenv('GEMHIP_EIG_TEST_BAIL_AFTER', raising=False)
        L.gemhip_set_host_threads(0, None)


def run_cases(check, rng):
    import numpy as np
    n = 300
    Q, _ = np.linalg.qr(rng.randn(n, n))
    lam = np.concatenate([1 + 1e-4 * rng.rand(200), 2 + rng.rand(68), [5, 5, 5, 5, 7, 7, 9, 9, 9, 9, 9, 9], np.zeros(20)])
    check((Q * lam) @ Q.T, 100)                                   # a bulk of width 1e-4 and exactly repeated values
    check(np.eye(150), 40)
    check(np.diag(np.arange(1.0, 201.0)), 60)
    A = np.zeros((160, 160)); A[:80, :80] = 3 * np.eye(80)
    check(A, 50)                                                   # decoupled tridiagonal, zero block
    B = rng.randn(100, 260)
    check(B.T @ B, 120)                                            # rank 100: the request reaches into the null space
    G = np.diag(np.concatenate([np.linspace(1, 0.3, 80), 0.2 * rng.rand(210)])); E = 1e-3 * rng.randn(290, 290)
    check(G + E + E.T, 80)                                         # what a restarted Rayleigh-Ritz matrix looks like


def test_gf_plan_rejects_edge_orders_the_reference_loop_cannot_be_scheduled_for():
    """gf.cpp:152-164 walks the file in order; (1,2),(0,1),(1,3) makes row 0 read row 1 between row 1's two updates, which the
    two-table schedule cannot express: GEMHIP_E_INVALID with a message, never a silently different result (validation only:
    the call returns before touching the device)."""
    import ctypes as C
    import numpy as np
    L = _hip.lib()
    src = np.array([1, 0, 1], np.int32); dst = np.array([2, 1, 3], np.int32)
    plan = C.c_void_p()
    rc = L.gemhip_gf_plan_create(4, 3, _hip.ptr(src, C.c_int32), _hip.ptr(dst, C.c_int32), None, 8, 0, 4, C.byref(plan))
    assert rc == -1 and not plan.value
    assert b'partly updated' in L.gemhip_last_error()
