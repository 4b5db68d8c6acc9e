#!/usr/bin/env python3
"""s_memtime phase split of sgns_win_kernel at SBM 1M/10M with r walks per node (GEMHIP_SGNS_PROFILE build: scripts/build_variant.sh prof -DGEMHIP_SGNS_PROFILE;
GEM_HIP_LIB=gem_amd/libgem_hip_prof.so python scripts/profile_sgns_phases.py [waves:prefetch:reload ...]).  The library prints the cycle sums per phase on stderr."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from gem_amd import _hip
from gem_amd.graph import edge_arrays, sbm_graph
from test_n2v_gpu import Dev
R = int(os.environ.get('DIAG_WALKS', '2'))
g = sbm_graph(1000000, 10000000, 100, 20260923 + 4)
n, src, dst, w, _ = edge_arrays(g)
dev = Dev(n, src, dst, w)
m = C.c_int64(); _hip.check(dev.L.gemhip_n2v_start_nodes(dev.h, C.byref(m)))
_hip.check(dev.L.gemhip_n2v_walks(dev.h, 1.0, 1.0, R, 80, 1, 11, 0, m.value * R, None))
dev.unigram()
tot = m.value * R * 80
for cs in sys.argv[1:] or ['1536:2:1']:
    waves, pf, rl = (int(x) for x in cs.split(':'))
    _hip.check(dev.L.gemhip_sgns_set_hogwild(dev.h, pf, rl))
    _hip.check(dev.L.gemhip_n2v_set_max_waves(dev.h, waves))
    _hip.check(dev.L.gemhip_sgns_init(dev.h, 128, 1, None, None))
    _hip.check(dev.L.gemhip_synchronize(None))
    pairs = C.c_int64(); _hip.check(dev.L.gemhip_sgns_pairs(dev.h, C.byref(pairs), 1))
    t = time.time()
    _hip.check(dev.L.gemhip_sgns_train(dev.h, 10, 5, 0.025, 1, 0, 0, m.value * R, tot, 0, 1, 11, None))
    _hip.check(dev.L.gemhip_synchronize(None))
    el = time.time() - t
    _hip.check(dev.L.gemhip_sgns_pairs(dev.h, C.byref(pairs), 0))
    print(json.dumps(dict(cfg=cs, seconds=round(el, 3), Mpairs_per_s=round(pairs.value / el / 1e6, 1), centres=int(tot))), flush=True)
dev.close()
