# ARCHIVED (round 3): drives the two-wavefront / global-scratch variants (GEMHIP_SGNS_DUO, GEMHIP_SGNS_OSCR), which was removed from the library; kept because committed profiles were produced with it
"""A/B of the LDS-window SGNS kernel against the round-1 kernel on the bench graph (SBM 1M/10M, d=128, r walks per node):
time per launch, algorithmic TB/s, and the reconstruction MAP over a fixed 256-node sample for every variant.
    python scripts/ab_sgns_window.py [r] [nodes] [edges] [blocks]
Variants: (label, flags, radius, delta, max_waves)."""
import os, sys, time, json, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from gem_amd import _hip, multi_gpu
from gem_amd.graph import sbm_graph, edge_arrays, to_csr
from gem_amd.evaluation import reconstruction as gr
r = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nodes = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
edges = int(sys.argv[3]) if len(sys.argv) > 3 else 10000000
blocks = int(sys.argv[4]) if len(sys.argv) > 4 else 100
variants = [('win_R10_delta_oscr', 11, 10, 1, 0, 0, '1'), ('r1_kernel', 11 | 128, 0, -1, 0), ('duo_delta', 11, 10, 1, 0, 1), ('duo_overwrite', 11, 10, 0, 0, 1), ('win_R10_delta', 11, 10, 1, 0), ('win_R10_overwrite', 11, 10, 0, 0),
            ('win_R7_delta', 11, 7, 1, 0), ('win_R5_delta', 11, 5, 1, 0), ('win_R10_delta_w1024', 11, 10, 1, 1024), ('win_w400', 11, 10, 1, 400), ('win_w800', 11, 10, 1, 800), ('win_w1536', 11, 10, 1, 1536), ('r1_w800', 11 | 128, 0, -1, 800)]
if len(sys.argv) > 5:
    keep = sys.argv[5].split(',')
    extra = [('win_w%d' % int(k[5:]), 11, 10, 1, int(k[5:])) for k in keep if k.startswith('win_w') and k not in [v[0] for v in variants]]
    variants = [v for v in variants + extra if v[0] in keep]
g = sbm_graph(nodes, edges, blocks, seed=20260923 + 4)
n, src, dst, w, _ = edge_arrays(g)
row_ptr, col, ww = to_csr(n, src, dst, w)
b = multi_gpu.HipBackendN2V(n, row_ptr, col, ww, 128)
L = _hip.lib()
m = b.num_start_nodes(); b.walks(1.0, 1.0, r, 80, 20260923, 11, 0, m * r); b.vocab(); b.build_unigram()
sample = np.random.RandomState(0).choice(n, size=min(1024, n), replace=False)
out = []
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 2
for rep in range(reps):
    for v in variants:
        name, flags, R, delta, mw = v[:5]
        os.environ['GEMHIP_SGNS_DUO'] = '1' if (len(v) > 5 and v[5]) else '0'
        os.environ['GEMHIP_SGNS_OSCR'] = v[6] if len(v) > 6 else '0'
        _hip.check(L.gemhip_sgns_set_window_cache(b.h, R if R > 0 else -1, delta))
        _hip.check(L.gemhip_n2v_set_max_waves(b.h, mw))
        b.init_tables(20260923); b.pairs(reset=True)
        torch.cuda.synchronize(); t = time.time()
        b.train(10, 1, 0, 0, m * r, m * r * 80, 0, 20260923, flags)
        torch.cuda.synchronize(); el = time.time() - t
        pairs = b.pairs()
        rec = {'variant': name, 'rep': rep, 'sgns_s': el, 'algorithmic_TBps': pairs * 7192 / el / 1e12, 'pairs': pairs,
               'finite': bool(torch.isfinite(b.P).all())}
        if rep == 1:
            rec['sampled_map'] = float(gr.sampled_ap_gpu(g, None, b.P.cpu().numpy(), sample).mean())
        print(json.dumps(rec), flush=True)
        out.append(rec)
json.dump(out, open('/root/repo/gpurun_out/ab_sgns_window_r%d_n%dk.json' % (r, nodes // 1000), 'w'), indent=1)
