"""ctypes binding of libgem_hip.so (the C ABI declared in include/gem_hip.h).

This is the ONLY way the Python layer reaches the device.  There is no CPU
fallback: if the shared library is missing or no MI355X is visible, calls
raise -- the product path never routes through oracle/.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('GEM_HIP_LIB') or os.path.join(_HERE, 'libgem_hip.so')     # GEM_HIP_LIB: an A/B or profiling build of the same ABI

_lib = None

i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)


class GemHipError(RuntimeError):
    pass


_SIGS = {
    'gemhip_version': (C.c_int, []),
    'gemhip_last_error': (C.c_char_p, []),
    'gemhip_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'gemhip_set_device': (C.c_int, [C.c_int]),
    'gemhip_malloc': (C.c_int, [C.POINTER(C.c_void_p), C.c_int64]),
    'gemhip_free': (C.c_int, [C.c_void_p]),
    'gemhip_memcpy_h2d': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    'gemhip_memcpy_d2h': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    'gemhip_synchronize': (C.c_int, [C.c_void_p]),
    'gemhip_last_call_phases': (C.c_int, [f64p]),
    'gemhip_gf_train': (C.c_int, [C.c_int64, C.c_int64, i32p, i32p, f32p, C.c_int32, C.c_float, C.c_float, C.c_int32,
                                  f32p, f64p]),
    'gemhip_gf_train_multi': (C.c_int, [C.c_int64, C.c_int64, i32p, i32p, f32p, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_int32, i32p, f32p, f64p]),
    'gemhip_n2v_train_multi': (C.c_int, [C.c_int64, C.c_int64, i64p, i32p, f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                         C.c_uint64, C.c_int32, C.c_int32, i32p, C.c_int32, f32p, f64p]),
    'gemhip_rccl_selftest': (C.c_int, [C.c_int32, i32p, C.c_int64, f64p]),
    'gemhip_gf_plan_create': (C.c_int, [C.c_int64, C.c_int64, i32p, i32p, f32p, C.c_int32, C.c_int64, C.c_int64,
                                        C.POINTER(C.c_void_p)]),
    'gemhip_gf_plan_destroy': (C.c_int, [C.c_void_p]),
    'gemhip_gf_plan_bind': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'gemhip_gf_plan_set_embedding': (C.c_int, [C.c_void_p, f32p]),
    'gemhip_gf_plan_init_embedding': (C.c_int, [C.c_void_p, C.c_uint64, C.c_float]),
    'gemhip_gf_plan_sweeps': (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_void_p]),
    'gemhip_gf_plan_set_rows_per_wave': (C.c_int, [C.c_void_p, C.c_int32]),
    'gemhip_gf_plan_set_fused_sweeps': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    'gemhip_gf_plan_get_embedding': (C.c_int, [C.c_void_p, f32p]),
    'gemhip_gf_plan_current': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    'gemhip_gf_plan_info': (C.c_int, [C.c_void_p, i64p]),
    'gemhip_gf_objective': (C.c_int, [C.c_int64, C.c_int64, i32p, i32p, f32p, C.c_int32, f32p, f64p]),
    'gemhip_hope_plan_create': (C.c_int, [C.c_int64, C.c_int64, i64p, i32p, f32p, C.c_float, C.POINTER(C.c_void_p)]),
    'gemhip_hope_plan_solve': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_uint64, f32p, f32p, f32p, f64p]),
    'gemhip_hope_plan_solve_device': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_uint64, C.c_void_p, C.c_void_p, f32p, f64p]),
    'gemhip_hope_plan_destroy': (C.c_int, [C.c_void_p]),
    'gemhip_hope': (C.c_int, [C.c_int64, C.c_int64, i64p, i32p, f32p, C.c_float, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                              C.c_float, C.c_uint64, f32p, f32p, f32p, f64p]),
    'gemhip_hope_svd_error': (C.c_int, [C.c_int64, C.c_int64, i64p, i32p, f32p, C.c_float, C.c_int32, f32p, f32p, C.c_int32, C.c_uint64, f64p, f64p]),
    'gemhip_hope_svd_error_uv': (C.c_int, [C.c_int64, C.c_int64, i64p, i32p, f32p, C.c_float, C.c_int32, f32p, f32p, f32p, C.c_int32, C.c_uint64, f64p, f64p, f64p]),
    'gemhip_hope_plan_svd_error_uv': (C.c_int, [C.c_void_p, C.c_int32, f32p, f32p, f32p, C.c_int32, C.c_uint64, f64p, f64p, f64p]),
    'gemhip_hope_plan_svd_error': (C.c_int, [C.c_void_p, C.c_int32, f32p, f32p, C.c_int32, C.c_uint64, f64p, f64p]),
    'gemhip_lap_eigmap': (C.c_int, [C.c_int64, C.c_int64, i64p, i32p, f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                    C.c_uint64, f32p, f32p, f64p]),
    'gemhip_lle': (C.c_int, [C.c_int64, C.c_int64, i64p, i32p, f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                             C.c_uint64, f32p, f32p, f64p]),
    'gemhip_sym_eig': (C.c_int, [C.c_int32, f64p, f64p]),
    'gemhip_sym_eig_builtin': (C.c_int, [C.c_int32, f64p, f64p]),
    'gemhip_sym_eig_top': (C.c_int, [C.c_int32, f64p, C.c_int32, f64p, f64p]),
    'gemhip_set_sym_eig_callback': (C.c_int, [C.c_void_p]),
    'gemhip_set_host_threads': (C.c_int, [C.c_int32, C.POINTER(C.c_int32)]),
    'gemhip_hope_spmm': (C.c_int, [C.c_int64, C.c_int64, i64p, i32p, f32p, C.c_float, C.c_int32, f32p, f32p, f32p]),
    'gemhip_hope_gram': (C.c_int, [C.c_int64, C.c_int32, C.c_int32, f32p, f32p, f64p]),
    'gemhip_hope_tsgemm': (C.c_int, [C.c_int64, C.c_int32, C.c_int32, f32p, f64p, C.c_float, f32p, f32p]),
    'gemhip_eval_sampled_ap': (C.c_int, [C.c_int64, C.c_int32, C.c_int32, f32p, f32p, i64p, i32p, C.c_int32, C.c_int64, i32p, f64p]),
    'gemhip_n2v_train': (C.c_int, [C.c_int64, C.c_int64, i64p, i32p, f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_float, C.c_float, C.c_uint64, C.c_int32, f32p, f64p]),
    'gemhip_n2v_create': (C.c_int, [C.c_int64, C.c_int64, i64p, i32p, f32p, C.POINTER(C.c_void_p)]),
    'gemhip_n2v_destroy': (C.c_int, [C.c_void_p]),
    'gemhip_n2v_build_alias': (C.c_int, [C.c_void_p, C.c_void_p]),
    'gemhip_n2v_get_alias': (C.c_int, [C.c_void_p, f32p, i32p, i32p]),
    'gemhip_n2v_walks': (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_uint64, C.c_int32, C.c_int64,
                                   C.c_int64, C.c_void_p]),
    'gemhip_n2v_start_nodes': (C.c_int, [C.c_void_p, i64p]),
    'gemhip_n2v_set_walks': (C.c_int, [C.c_void_p, i32p, C.c_int64, C.c_int32, C.c_int64]),
    'gemhip_n2v_get_walks': (C.c_int, [C.c_void_p, i32p]),
    'gemhip_n2v_walks_ptr': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), i64p, i32p]),
    'gemhip_n2v_vocab': (C.c_int, [C.c_void_p, C.c_void_p]),
    'gemhip_n2v_counts_ptr': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    'gemhip_n2v_build_unigram_vocab_order': (C.c_int, [C.c_void_p, C.c_int32, i64p, i32p, f32p, i32p]),
    'gemhip_n2v_build_unigram': (C.c_int, [C.c_void_p, i32p, f32p, i32p]),
    'gemhip_sgns_init': (C.c_int, [C.c_void_p, C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p]),
    'gemhip_n2v_bind_counts': (C.c_int, [C.c_void_p, C.c_void_p]),
    'gemhip_sgns_pairs': (C.c_int, [C.c_void_p, i64p, C.c_int32]),
    'gemhip_n2v_build_unigram_parts': (C.c_int, [C.c_void_p, C.c_int32, f32p, i32p]),
    'gemhip_n2v_build_unigram_parts_vocab_order': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, f32p, i32p, i32p, i64p]),
    'gemhip_sgns_train_part': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_float,
                                         C.c_int64, C.c_int64, C.c_int32, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_void_p]),
    'gemhip_n2v_copy_walks': (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    'gemhip_n2v_set_max_waves': (C.c_int, [C.c_void_p, C.c_int32]),
    'gemhip_sgns_set_window_cache': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    'gemhip_sgns_set_hogwild': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    'gemhip_sgns_set_hot_rows': (C.c_int, [C.c_void_p, C.c_int32]),
    'gemhip_sgns_last_launch': (C.c_int, [C.c_void_p, i32p, i32p, i32p, i32p]),
    'gemhip_sgns_set_fresh': (C.c_int, [C.c_void_p, C.c_int32]),
    'gemhip_n2v_locally_hot': (C.c_int, [C.c_void_p, C.c_int32, i64p, i32p]),
    'gemhip_n2v_locally_hot_corpus': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, i64p, C.c_void_p]),
    'gemhip_sgns_plan_launch': (C.c_int, [i32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, i32p, i32p, i32p, f64p, f64p]),
    'gemhip_test_wave_sum6': (C.c_int, [f32p, f32p]),
    'gemhip_sgns_set_tables': (C.c_int, [C.c_void_p, f32p, f32p]),
    'gemhip_sgns_get_tables': (C.c_int, [C.c_void_p, f32p, f32p]),
    'gemhip_sgns_train': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32, C.c_int64, C.c_int64,
                                    C.c_int64, C.c_int64, C.c_uint64, C.c_int32, C.c_void_p]),
}


def declared_symbols():
    """Names this binding expects; tests check them against include/gem_hip.h."""
    return sorted(_SIGS)


def lib():
    """Load libgem_hip.so (once).  Raises GemHipError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GemHipError('libgem_hip.so not built: run `python -m gem_amd.build` (needs hipcc). '
                              'There is no CPU fallback.')
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)      # AttributeError here == ABI drift; let it propagate
            fn.restype = res
            fn.argtypes = args
        _lib = L
        if os.environ.get('GEM_HIP_LAPACK_EIG') == '1':
            _register_lapack(L)
    return _lib


_EIG_CB_TYPE = C.CFUNCTYPE(C.c_int, C.c_int32, f64p, f64p)
_eig_cb_keepalive = None


def _register_lapack(L):
    """Hand numpy's LAPACK symmetric eigensolver to the library for HOPE's projected (<= 512 x 512) problems.
    OFF by default: on the 256-core GPU host OpenBLAS spins up every core for a 200 x 200 problem and is 1.4x
    SLOWER than the built-in single-threaded Householder/QL (measured: 20 ms vs 15 ms per solve).  GEM_HIP_LAPACK_EIG=1."""
    global _eig_cb_keepalive

    def _eigh(n, a_ptr, w_ptr):
        try:
            A = np.ctypeslib.as_array(a_ptr, shape=(n, n))
            w, V = np.linalg.eigh(A)
            A[:, :] = V
            np.ctypeslib.as_array(w_ptr, shape=(n,))[:] = w
            return 0
        except Exception:           # fall back to the built-in solver
            return 1
    _eig_cb_keepalive = _EIG_CB_TYPE(_eigh)
    L.gemhip_set_sym_eig_callback(C.cast(_eig_cb_keepalive, C.c_void_p))


def check(rc):
    if rc != 0:
        raise GemHipError('libgem_hip error %d: %s' % (rc, lib().gemhip_last_error().decode('utf-8', 'replace')))


def device_count():
    n = C.c_int(0)
    rc = lib().gemhip_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def require_device():
    if device_count() < 1:
        raise GemHipError('no HIP device visible: the gem_amd backend needs an MI355X (gfx950); '
                          'there is no CPU fallback (%s)' % lib().gemhip_last_error().decode())


def ptr(a, ctype):
    return None if a is None else a.ctypes.data_as(C.POINTER(ctype))


E_INVALID, E_HIP, E_UNSUPPORTED, E_NOTCONVERGED = -1, -2, -3, -4          # include/gem_hip.h GEMHIP_E_*
N2V_PAD_ZERO, N2V_UNIGRAM_QUIRK, N2V_DETERMINISTIC, N2V_UNIFORM_FIRST_HOP = 1, 2, 4, 8
N2V_SNAP_COMPAT = 11
N2V_VOCAB_ORDER = 16              # unigram table in the binary's layout (first-appearance order over the nodes that occur)
N2V_SNAP_LAYOUT = 27             # SNAP_COMPAT | VOCAB_ORDER: what node2vec.learn_embedding passes by default
N2V_NO_WINDOW_CACHE = 128       # A/B switch: round-1 SGNS kernel without the LDS window of context rows


def as_i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def as_f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def warn_if_unconverged(stats, tol, max_restarts, what):
    """The block-Krylov solver returns its best Ritz pairs when the restart budget runs out; scipy's svds / eigs would raise
    ArpackNoConvergence there.  Keep GEM's "returns an embedding" behaviour but say so."""
    import warnings
    if stats.get('restarts', 0) >= max_restarts and stats.get('last_sigma_change', 0.0) >= tol:
        warnings.warn('%s: not converged after %d restarts (last relative change of the wanted values %.2e >= tol %.1e, Ritz residual %.2e); '
                      'raise max_restarts or tol' % (what, int(stats['restarts']), stats['last_sigma_change'], tol, stats.get('ritz_residual', float('nan'))),
                      RuntimeWarning, stacklevel=3)


def last_call_phases():
    """{total, host_prepare, h2d, kernels, d2h} seconds of the last one-shot library call on this thread (gemhip_last_call_phases)."""
    out = (C.c_double * 8)()
    check(lib().gemhip_last_call_phases(out))
    return {'library_call_s': out[0], 'host_prepare_s': out[1], 'h2d_s': out[2], 'kernels_s': out[3], 'd2h_s': out[4]}


def api_wall(t_begin, t_ingested, t_called, t_end):
    """SURVEY 8(d) "API wall" of one learn_embedding(): time.perf_counter() stamps at entry, after the graph became arrays (ingest),
    after the library call, at return (float64 copy done) -> the breakdown bench.py prints as `api_wall`."""
    ph = last_call_phases()
    return {'seconds': t_end - t_begin, 'ingest_s': t_ingested - t_begin, 'host_prepare_s': ph['host_prepare_s'], 'h2d_s': ph['h2d_s'],
            'kernels_s': ph['kernels_s'], 'd2h_s': ph['d2h_s'], 'd2h_float64_s': ph['d2h_s'] + (t_end - t_called),
            'float64_copy_s': t_end - t_called, 'library_call_s': t_called - t_ingested}
