"""A/B of GF's launch loop against the cooperative multi-sweep launch (gf_sweeps_coop_kernel, gemhip_gf_plan_set_fused_sweeps) at BASELINE configs[1]
(SBM 10k/100k, d=128, run_sbm.py's eta/lambda): microseconds per sweep over 1000 sweeps, tables bit-identical.  One JSON line per setting.
    python scripts/ab_gf_fused.py [sweeps_per_launch[/max_grid] ...]        (0 = the launch loop)"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from gem_amd import _hip
from gem_amd.graph import sbm_graph, edge_arrays

specs = sys.argv[1:] or ['0', '8', '64', '1000', '1000/2048', '1000/1024', '1000/512', '1000/256']
nodes, edges, blocks = (int(os.environ.get(k, v)) for k, v in (('NODES', 10000), ('EDGES', 100000), ('BLOCKS', 10)))
g = sbm_graph(nodes, edges, blocks, seed=20260923 + 4)
n, src, dst, w, _ = edge_arrays(g)
d, L = 128, _hip.lib()
dev = torch.device('cuda', 0)
X0 = (0.01 * torch.randn(n, d, device=dev, generator=torch.Generator(device=dev).manual_seed(1234))).contiguous()
ref = None
for spec in specs:
    k, grid = (int(x) for x in (spec.split('/') + ['0'])[:2])
    Xa, Xb = X0.clone(), X0.clone()
    plan = C.c_void_p()
    _hip.check(L.gemhip_gf_plan_create(n, len(src), _hip.ptr(src, C.c_int32), _hip.ptr(dst, C.c_int32), None, d, 0, n, C.byref(plan)))
    _hip.check(L.gemhip_gf_plan_bind(plan, C.c_void_p(Xa.data_ptr()), C.c_void_p(Xb.data_ptr())))
    _hip.check(L.gemhip_gf_plan_set_fused_sweeps(plan, k, grid))
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _hip.check(L.gemhip_gf_plan_sweeps(plan, 100, 1e-4, 1.0, s))                 # warm-up
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); _hip.check(L.gemhip_gf_plan_sweeps(plan, 1000, 1e-4, 1.0, s)); e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 1000
    cur = C.c_void_p(); _hip.check(L.gemhip_gf_plan_current(plan, C.byref(cur)))
    X = Xa if cur.value == Xa.data_ptr() else Xb
    if ref is None:
        ref = X.clone()
    print(json.dumps(dict(sweeps_per_launch=k, max_grid=grid, us_per_sweep=us, edges_per_s=g.number_of_edges() / us * 1e6, bit_identical_to_first=bool(torch.equal(X, ref)))), flush=True)
    _hip.check(L.gemhip_gf_plan_destroy(plan))
