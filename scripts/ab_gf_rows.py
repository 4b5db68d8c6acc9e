"""A/B of the GF sweep kernels at SBM 1M/10M (d=128): rows per wavefront 1 (gf_sweep_kernel) against K > 1 (gf_sweep_rows_kernel); the tables
must be bit-identical.  One JSON line per setting:  python scripts/ab_gf_rows.py [K | K/NT ...]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from gem_amd import _hip
from gem_amd.graph import sbm_graph, edge_arrays

ks = [a for a in sys.argv[1:]] or ['1', '2', '4', '8', '16', '32']
nodes, edges, blocks = (int(os.environ.get(k, v)) for k, v in (('NODES', 1000000), ('EDGES', 10000000), ('BLOCKS', 100)))
g = sbm_graph(nodes, edges, blocks, seed=20260923 + 4)
n, src, dst, w, _ = edge_arrays(g)
d, L = 128, _hip.lib()
dev = torch.device('cuda', 0)
X0 = (0.01 * torch.randn(n, d, device=dev, generator=torch.Generator(device=dev).manual_seed(1234))).contiguous()
ref = None
for spec in ks:
    k, nt = (int(x) for x in (spec.split('/') + ['0'])[:2])          # "K" or "K/NT": NT = GEMHIP_GF_NT_STORE of the plan (1: non-temporal row stores, 2: own-row loads, 3: both)
    os.environ['GEMHIP_GF_NT_STORE'] = str(nt)
    Xa, Xb = X0.clone(), X0.clone()
    plan = C.c_void_p()
    _hip.check(L.gemhip_gf_plan_create(n, len(src), _hip.ptr(src, C.c_int32), _hip.ptr(dst, C.c_int32), None, d, 0, n, C.byref(plan)))
    _hip.check(L.gemhip_gf_plan_bind(plan, C.c_void_p(Xa.data_ptr()), C.c_void_p(Xb.data_ptr())))
    _hip.check(L.gemhip_gf_plan_set_rows_per_wave(plan, k))
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _hip.check(L.gemhip_gf_plan_sweeps(plan, 10, 1e-2, 1e-2, s))                 # warm-up (even count: result lands in Xa)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); _hip.check(L.gemhip_gf_plan_sweeps(plan, 50, 1e-2, 1e-2, s)); e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    info = (C.c_int64 * 8)(); _hip.check(L.gemhip_gf_plan_info(plan, info))
    comp = info[1] * 2 * 4 * d + info[0] * (4 * d + 8)
    if ref is None:
        ref = Xa.clone()
    print(json.dumps(dict(rows_per_wave=k, nt_store=nt, us_per_sweep=us, compulsory_GBs=comp / us / 1e3, frac_of_8TBs=comp / us / 1e3 / 8000.0,
                          bit_identical_to_first=bool(torch.equal(Xa, ref)), rows=info[1], updates=info[0])), flush=True)
    _hip.check(L.gemhip_gf_plan_destroy(plan))
