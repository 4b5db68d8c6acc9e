#!/usr/bin/env python3
"""Pair the per-node APs that scripts/check_rmat17_launches.py saved on the GPU box with the sequential oracle's APs of the same nodes
(tests/golden/n2v_ref_oracle_rmat17*_e16k.json: the `big` sample = reconstruction.eligible_sample(g, 16384)) and write one JSON line per launch:
the paired gap in % of the oracle's MAP and its standard error -- next to the old 2 048-node uniform-sample statistic of rounds 2-4, for the record.

    python scripts/pair_rmat_launches.py gpurun_out/r05_rmat17 _b [box [scale]] > profiles/r05_rmat17_width_sweep.jsonl
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import numpy as np
d, tag = sys.argv[1], sys.argv[2]
box = sys.argv[3] if len(sys.argv) > 3 else tag.strip('_')
scale = int(sys.argv[4]) if len(sys.argv) > 4 else 17
G = os.path.join(ROOT, 'tests', 'golden')


def _load(name):
    p = os.path.join(G, name)
    return json.load(open(p)) if os.path.exists(p) else None


# `big` = the eligible-sample goldens; `old` = the 2 048-node uniform-sample goldens of rounds 2-4 (scale 17 only)
ref = {11: {'big': _load('n2v_ref_oracle_rmat%d_e16k.json' % scale), 'old': _load('n2v_ref_oracle_rmat%d.json' % scale)},
       27: {'big': _load('n2v_ref_oracle_rmat%d_vocab_order_e16k.json' % scale), 'old': _load('n2v_ref_oracle_rmat%d_vocab_order.json' % scale)}}


def gap(ap, r):
    dd = ap - np.asarray(r['ap'])
    return round(float(100 * dd.mean() / r['MAP']), 3), round(float(100 * dd.std(ddof=1) / np.sqrt(len(dd)) / r['MAP']), 3)


recs = [json.loads(l) for l in open(os.path.join(d, 'launches%s.jsonl' % tag))]
idx = {}
for r in recs:
    if r.get('mode') == 'plan':
        print(json.dumps(dict(r, box=box)))
    if r.get('mode') not in ('one_shot', 'staged'):
        continue
    fl = r['flags']
    kind = 'f' if r['mode'] == 'one_shot' else 'sweep_f'
    k = idx.get((kind, fl), 0); idx[(kind, fl)] = k + 1
    out = {'box': box, 'mode': r['mode'], 'flags': fl, 'sgns_s': r['sgns_s']}
    for key in ('launch', 'max_waves', 'hot_count', 'rep'):
        if key in r:
            out[key] = r[key]
    for s in ('big', 'old'):
        if ref[fl][s] is None:
            continue
        A = np.load(os.path.join(d, 'ap_%s_%s%d%s.npy' % (s, kind, fl, tag)))
        g_, se = gap(A[k].astype(np.float64), ref[fl][s])
        out['gap_%s_pct' % s] = g_; out['gap_%s_se_pct' % s] = se
    print(json.dumps(out))
