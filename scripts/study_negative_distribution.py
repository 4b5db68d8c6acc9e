#!/usr/bin/env python3
"""The negative-sampling distribution TrainModel ACTUALLY draws from under RndUnigramInt's quirk, per unigram-table layout, on the R-MAT graphs of
tests/test_rmat_gpu.py -- and the statistics the Hogwild launch rule (plan_sgns_launch, n2v.hip) would see if it used it instead of the nominal
unigram^0.75 distribution.  CPU only (oracle tables = the library's, bit for bit: tests/test_n2v_gpu.py).

The quirk (SNAP's RndUnigramInt as the binary calls it): the uniform slot names X = KTable[slot], and the draw is X with probability UTable[X], else
KTable[X] -- so only ALIAS TARGETS are ever drawn, with masses that depend on Vose's pairing, i.e. on the order of the table:
    P(X = x) = #{slot: KTable[slot] = x} / slots,      q(v) = sum_x P(x) (UT[x] [x = v] + (1 - UT[x]) [KT[x] = v]).
Node-id layout (flags 11): slots = all n nodes (those that never occur have weight 0 and alias to a large node); the binary's layout (flags 27): slots =
the nodes that occur, in order of first appearance.

    python scripts/study_negative_distribution.py --out profiles/r05_negative_distribution_by_layout.json
"""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def actual_q(n, slot_tab, UT, KT):
    PX = np.bincount(slot_tab, minlength=n) / float(len(slot_tab))
    q = PX * UT
    np.add.at(q, KT, PX * (1.0 - UT))
    return q


def study(scale, edges):
    import oracle
    from gem_amd import _hip
    from gem_amd.graph import edge_arrays, rmat_graph
    g = rmat_graph(scale, edges, 20260928)
    n, src, dst, _, _ = edge_arrays(g)
    row_ptr, col, _ = oracle.sorted_csr(n, src, dst, None)
    t = time.time()
    walks = oracle.n2v_walks(row_ptr, col, None, None, 1.0, 1.0, 10, 80, 20260923, 11)
    cnt = oracle.n2v_vocab(n, walks)
    tot = float(cnt.sum()); p = cnt / tot
    z = cnt.astype(np.float64) ** 0.75; q_nom = z / z.sum()
    UT, KT = oracle.unigram_build(cnt)
    q11 = actual_q(n, KT.astype(np.int64), UT.astype(np.float64), KT.astype(np.int64))
    slot_tab, UTn, KTn = oracle.unigram_build_vocab_order(cnt, walks, 27)[:3]
    q27 = actual_q(n, slot_tab.astype(np.int64), UTn.astype(np.float64), KTn.astype(np.int64))
    active = int((cnt > 0).sum())
    rec = {'graph': 'R-MAT scale %d' % scale, 'n': int(n), 'active_nodes': active, 'walks': int(walks.shape[0]), 'tokens': int(tot),
           'largest_token_share': float(p.max()), 'seconds': round(time.time() - t, 1), 'layouts': {}}
    L = _hip.lib()
    for name, fl, q in (('nominal unigram^0.75', 27, q_nom), ('node_id (flags 11), as sampled', 11, q11), ('vocab_order (flags 27), as sampled', 27, q27)):
        k_, w_, hot_, ne_, nec_ = C.c_int32(), C.c_int32(), C.c_int32(), C.c_double(), C.c_double()
        _hip.check(L.gemhip_sgns_plan_launch(_hip.ptr(cnt, C.c_int32), n, 128, 10, 80, walks.shape[0], fl, C.byref(k_), C.byref(w_), C.byref(hot_), C.byref(ne_), C.byref(nec_)))
        load = p + 5.0 * q
        t2 = float((load ** 2).sum()); hub = max(0.0, t2 - 40.0 / active)
        cold = cnt < hot_.value
        s1, s2 = float(load[cold].sum()), float((load[cold] ** 2).sum())
        rec['layouts'][name] = {
            'nodes_ever_drawn': int((q > 0).sum()), 'sum_q2': float((q ** 2).sum()), 'largest_q': float(q.max()), 'largest_row_load_p_plus_5q': float(load.max()),
            'touch2': t2, 'touch2_hub': hub, 'planned_wavefronts': int(w_.value), 'hot_threshold_tokens': int(hot_.value),
            'cold_rows_collision_n_eff': s1 * s1 / s2, 'largest_cold_row_load': float(load[cold].max()),
            'cold_rows_drawn_over_10x_nominal': int(((q > 10 * q_nom) & cold & (cnt > 0)).sum()),
            'negative_mass_on_those': float(q[(q > 10 * q_nom) & cold & (cnt > 0)].sum())}
    return rec


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--graphs', default='17:2000000,20:16000000')
    ap.add_argument('--out', required=True)
    a = ap.parse_args()
    out = {'measured_gap_to_the_sequential_oracle_pct': {
        'scale 17': {'vocab_order': {'128': -0.1, '256': -0.9, '512': -4.2, '768': -6.3, '50': -0.33}, 'node_id': {'128': -0.2, '256': 0.2, '512': -1.7, '768': -2.4, '50': -0.73}},
        'scale 20': {'vocab_order': {'688': -6.4, '256': -1.9, '207': -1.51}, 'node_id': {'688': -8.3, '207': -6.29, '104': -1.50}},
        'source': 'profiles/r05_rmat17_width_sweep.jsonl, r05_rmat20_launches_e128k.jsonl, r05_pytest_gpu_final2_scale20_*.log (keys: concurrent wavefronts)'},
        'graphs': []}
    for spec in a.graphs.split(','):
        s, e = spec.split(':')
        out['graphs'].append(study(int(s), int(e)))
        json.dump(out, open(a.out, 'w'), indent=1)
    print(json.dumps(out, indent=1))
