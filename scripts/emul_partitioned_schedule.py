"""CPU emulation (oracle only) of the partitioned walk-ordered schedule: MAP of parts x parts buckets per episode against the sequential TrainModel
on the same walks and draws (paired per node).  usage: emul_partitioned_schedule.py nodes walks_per_node d parts episodes [parts episodes ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import oracle
from gem_amd.graph import sbm_graph, edge_arrays
from gem_amd.evaluation import reconstruction as gr

nodes, r, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfgs = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(4, len(sys.argv), 2)]
g = sbm_graph(nodes, nodes * 10, max(4, nodes // 10000), seed=11)
n, src, dst, w, _ = edge_arrays(g)
rp, cs, _ = oracle.sorted_csr(n, src, dst, None)
L, win, seed, flags = 80, 10, 5, 11
walks = oracle.n2v_walks(rp, cs, None, None, 1.0, 1.0, r, L, seed, flags)
cnt = oracle.n2v_vocab(n, walks)
tot = walks.size
sample = np.random.RandomState(0).choice(n, size=min(n, 2048), replace=False)


def ap_of(P):
    P = P.astype(np.float64)
    return gr.sampled_map(g, lambda i: P @ P[i], sample)


UT, KT = oracle.unigram_build(cnt)
t = time.time()
P0, N0 = oracle.sgns_init(n, d, seed)
oracle.sgns_train(walks, win, 0.025, 1, 0, tot, 0, 0, UT, KT, seed, flags, P0, N0)
m0 = ap_of(P0)
print(json.dumps(dict(schedule='sequential', seconds=time.time() - t, MAP=float(np.mean(m0)) if np.ndim(m0) else float(m0))), flush=True)
for parts, episodes in cfgs:
    t = time.time()
    UTp, KTp, off = oracle.unigram_build_parts(cnt, parts)
    P, N = oracle.sgns_init(n, d, seed)
    nw = walks.shape[0]
    done = 0
    for e in range(episodes):
        a, z = nw * e // episodes, nw * (e + 1) // episodes
        for s in range(parts):
            for gq in range(parts):
                h = (gq + s) % parts
                oracle.sgns_train_part(walks[a:z], None, win, 0.025, tot, done, 0, parts, gq, h, UTp[off[h]:off[h + 1]], KTp[off[h]:off[h + 1]], seed, flags,
                                       P, N, walk_id_offset=a)
        done += (z - a) * L
    m1 = ap_of(P)
    print(json.dumps(dict(schedule='partitioned', parts=parts, episodes=episodes, seconds=time.time() - t, MAP=float(np.mean(m1)) if np.ndim(m1) else float(m1),
                          vs_sequential_pct=100 * (float(np.mean(m1)) / float(np.mean(m0)) - 1))), flush=True)
