"""Margins of the statistical (Hogwild) GPU tests: prints measured MAPs next to their reference values and bars."""
import json, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from conftest import golden_path, load_sbm1024
from gem_amd.embedding.node2vec import node2vec
from gem_amd.evaluation import reconstruction as gr

G = load_sbm1024()
ref = json.load(open(golden_path('n2v_ref.json')))
for d, seeds, key, bar in ((16, (1, 2, 3), 'sbm1024_d16', 0.03), (128, (4, 5, 6), 'sbm1024_d128', 0.056)):
    for rep in range(3):
        maps = []
        for seed in seeds:
            m = node2vec(d=d, max_iter=1, walk_len=80, num_walks=10, con_size=10, ret_p=1, inout_p=1, seed=seed)
            Y = m.learn_embedding(graph=G, is_weighted=True, no_python=True)
            maps.append(gr.evaluateStaticGraphReconstruction(G, m, Y, None)[0])
        t1 = np.mean(ref[key + '_t1'])
        print('d=%d rep %d: maps %s mean %.4f  ref t1 %.4f  rel.dev %.2f%% (bar %.1f%%)  t8 %.4f' % (
            d, rep, np.round(maps, 4), np.mean(maps), t1, 100 * (np.mean(maps) - t1) / t1, 100 * bar, np.mean(ref[key + '_t8'])), flush=True)
