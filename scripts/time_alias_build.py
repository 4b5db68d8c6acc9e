#!/usr/bin/env python3
"""First-order alias tables (PreprocessTransitionProbs / GetNodeAlias) of a WEIGHTED power-law graph, timed: R-MAT (BASELINE configs[4] is scale 22,
4.2 M nodes / 62 M edges, max degree 94 115) with heavy-tailed edge weights.  The one-lane-per-row kernel walks a hub's stacks as one dependent
chain; rows of >= 2048 neighbours now take n2v_alias_hub_kernel (closed form, a workgroup per row).  GEMHIP_ALIAS_HUB_DEG is compiled in, so the
"before" figure comes from the baseline library (GEM_HIP_LIB).   python scripts/time_alias_build.py [scale] [edges]"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from gem_amd import _hip
from gem_amd.graph import rmat_graph, edge_arrays, to_csr

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
edges = int(sys.argv[2]) if len(sys.argv) > 2 else 64000000
t0 = time.time()
g = rmat_graph(scale, edges, seed=20260923)
n, src, dst, w, _ = edge_arrays(g)
rng = np.random.RandomState(1)
w = (rng.pareto(1.5, len(src)) + 0.05).astype(np.float32)
row_ptr, col, ww = to_csr(n, src, dst, w, sort_cols=True)
t_gen = time.time() - t0
i64p, i32p, f32p = C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_float)
deg = np.diff(row_ptr)
libs = [p for p in (os.path.join(ROOT, 'gem_amd', 'libgem_hip_r04base.so'), _hip.LIB_PATH) if os.path.exists(p)]     # (the baseline build, if it was kept: one lane per row for every row)
for path in libs:
    L = C.CDLL(path)
    L.gemhip_n2v_create.argtypes = [C.c_int64, C.c_int64, i64p, i32p, f32p, C.POINTER(C.c_void_p)]
    L.gemhip_n2v_build_alias.argtypes = [C.c_void_p, C.c_void_p]
    L.gemhip_n2v_destroy.argtypes = [C.c_void_p]
    L.gemhip_synchronize.argtypes = [C.c_void_p]
    h = C.c_void_p()
    assert L.gemhip_n2v_create(n, len(col), _hip.ptr(row_ptr, C.c_int64), _hip.ptr(col, C.c_int32), _hip.ptr(ww, C.c_float), C.byref(h)) == 0
    assert L.gemhip_synchronize(None) == 0
    t = time.time()
    assert L.gemhip_n2v_build_alias(h, None) == 0
    assert L.gemhip_synchronize(None) == 0
    el = time.time() - t
    print(json.dumps(dict(lib=os.path.basename(path), scale=scale, nodes=int(n), directed_edges=int(len(col)), max_degree=int(deg.max()),
                          rows_ge_2048=int((deg >= 2048).sum()), entries_in_those_rows=int(deg[deg >= 2048].sum()), alias_build_seconds=el,
                          graph_generation_seconds=t_gen)), flush=True)
    assert L.gemhip_n2v_destroy(h) == 0
