# ARCHIVED (round 3): drives the three-wavefront kernel (gemhip_sgns_set_team), which was removed from the library; kept because committed profiles were produced with it
#!/usr/bin/env python3
"""Timing diagnosis of sgns_team_kernel at SBM 1M/10M with r walks per node (default 2): the kernel with parts switched off
(GEMHIP_TEAM_DIAG build, flags >> 16: 1 no LDS adds, 2 no repeat barriers, 4 no boundary barriers / fold, 8 no row stores -- results are
wrong, only the time is of interest) and, with a GEMHIP_SGNS_PROFILE build, the s_memtime phase split per wavefront role.
    GEM_HIP_LIB=gem_amd/libgem_hip_diag.so python scripts/diag_team.py team:1024:1:0 team:1024:1:1 ... win:1024:0:0"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
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
for cs in sys.argv[1:]:
    kern, walks, pf, dbg = cs.split(':')
    _hip.check(dev.L.gemhip_sgns_set_team(dev.h, -1 if kern == 'team' else 0, int(pf)))
    _hip.check(dev.L.gemhip_n2v_set_max_waves(dev.h, int(walks)))
    _hip.check(dev.L.gemhip_sgns_init(dev.h, 128, 1, None, None))
    _hip.check(dev.L.gemhip_synchronize(None))
    pairs = C.c_int64(); _hip.check(dev.L.gemhip_sgns_pairs(dev.h, C.byref(pairs), 1))
    t = time.time()
    _hip.check(dev.L.gemhip_sgns_train(dev.h, 10, 5, 0.025, 1, 0, 0, m.value * R, tot, 0, 1, 11 | (int(dbg) << 16), None))
    _hip.check(dev.L.gemhip_synchronize(None))
    el = time.time() - t
    _hip.check(dev.L.gemhip_sgns_pairs(dev.h, C.byref(pairs), 0))
    print(json.dumps(dict(cfg=cs, seconds=round(el, 3), Mpairs_per_s=round(pairs.value / el / 1e6, 1), us_per_pair_per_walk=round(el * int(walks) / pairs.value * 1e6, 3))), flush=True)
dev.close()
