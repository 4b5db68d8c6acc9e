#!/usr/bin/env python3
"""HOPE at BASELINE configs[2] (SBM 100k/1M, k=64, beta=0.01): the symmetric eigen-path (Chebyshev-filtered subspace iteration on A)
against the general block-Krylov solver on S^T S -- seconds per solve, SpMM launches/columns, and the 64 singular values of each
against the ARPACK golden (tests/golden/hope_sigma_sbm100k.json).  One JSON line per variant.  Arguments, if any, replace the variants:
name:ENV=VAL,ENV=VAL (eigen-path forced on)."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gem_amd import _hip
from gem_amd.graph import sbm_graph, edge_arrays, to_csr

ref = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'hope_sigma_sbm100k.json')))
pr = ref['params']
g = sbm_graph(pr['n'], pr['edges'], pr['blocks'], pr['seed'])
n, src, dst, w, _ = edge_arrays(g)
row_ptr, col, _ = to_csr(n, src, dst, None)
k = 64
L = _hip.lib()
plan = C.c_void_p()
_hip.check(L.gemhip_hope_plan_create(n, len(col), _hip.ptr(row_ptr, C.c_int64), _hip.ptr(col, C.c_int32), None, 0.01, C.byref(plan)))
U = np.empty((n, k), np.float32); V = np.empty((n, k), np.float32); sig = np.empty(k, np.float32)
stats = (C.c_double * 12)()
variants = [('block_krylov', {'GEMHIP_HOPE_SYM': '0'}), ('sym', {'GEMHIP_HOPE_SYM': '1'})] + \
           [('sym_' + '_'.join('%s%s' % (a[16:].lower(), b) for a, b in sorted(e.items())), dict(e, GEMHIP_HOPE_SYM='1')) for e in (
               {'GEMHIP_HOPE_SYM_AMP': '1e5'}, {'GEMHIP_HOPE_SYM_MAXDEG': '30'})]
if sys.argv[1:]:      # python scripts/ab_hope_sym.py name:ENV=VAL,ENV=VAL ...   (e.g. fused:GEMHIP_HOPE_SYM_FUSED_RR=1 two_pass:GEMHIP_HOPE_SYM_FUSED_RR=0)
    variants = [(a.split(':', 1)[0], dict(GEMHIP_HOPE_SYM='1', **dict(kv.split('=', 1) for kv in a.split(':', 1)[1].split(',') if kv))) for a in sys.argv[1:]]
s_ref = np.asarray(ref['sigma_ascending'])
for name, env in variants:
    for kk in list(os.environ):
        if kk.startswith('GEMHIP_HOPE_') and kk != 'GEMHIP_HOPE_SPMM16': del os.environ[kk]      # SPMM16 (kernel choice) is read once per process
    os.environ.update(env)
    if name == 'sym': os.environ['GEMHIP_HOPE_DEBUG'] = '1'
    ts = []
    for rep in range(4):
        t = time.time()
        _hip.check(L.gemhip_hope_plan_solve(plan, k, 16, 3, 20, 1e-5, 20260923, _hip.ptr(U, C.c_float), _hip.ptr(V, C.c_float), _hip.ptr(sig, C.c_float), stats))
        ts.append(time.time() - t)
        os.environ.pop('GEMHIP_HOPE_DEBUG', None)
    Un = U / np.sqrt(sig); Vn = V / np.sqrt(sig)
    print(json.dumps({'variant': name, 'seconds_min': min(ts[1:]), 'seconds': ts, 'device_seconds': stats[0], 'spmm_launches': stats[1], 'spmm_columns': stats[2],
                      'katz_terms': stats[3], 'cycles': stats[5], 'spmm_seconds': stats[11], 'host_eig_seconds': stats[8], 'last_change': stats[6], 'residual': stats[10],
                      'max_rel_err_vs_arpack': float(np.abs(sig / s_ref - 1).max()),
                      'orth_err': float(max(np.abs(Un.T @ Un - np.eye(k)).max(), np.abs(Vn.T @ Vn - np.eye(k)).max()))}), flush=True)
L.gemhip_hope_plan_destroy(plan)
# how long the two n x k float32 downloads into pageable numpy memory take on their own (they are inside every solve above)
hip = C.CDLL('libamdhip64.so')
dptr = C.c_void_p()
assert hip.hipMalloc(C.byref(dptr), C.c_size_t(U.nbytes)) == 0
ts = []
for rep in range(4):
    t = time.time()
    assert hip.hipMemcpy(U.ctypes.data_as(C.c_void_p), dptr, C.c_size_t(U.nbytes), 2) == 0
    assert hip.hipMemcpy(V.ctypes.data_as(C.c_void_p), dptr, C.c_size_t(V.nbytes), 2) == 0
    ts.append(time.time() - t)
print(json.dumps({'variant': 'download_only_2x%dMB_pageable' % (U.nbytes >> 20), 'seconds_min': min(ts), 'seconds': ts}), flush=True)
hip.hipFree(dptr)
