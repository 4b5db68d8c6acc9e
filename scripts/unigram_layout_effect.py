"""Does the LAYOUT of the unigram alias table matter?  The reference binary renames tokens by first appearance in the walk matrix, so its
vocabulary -- and the alias table RndUnigramInt draws negatives from -- is laid out in that order; oracle/n2v_oracle.c and the HIP path keep
node-id order.  Under RndUnigramInt's quirk (X = KTable[slot]) the negative distribution is a function of the table's alias structure, hence
of the layout.  This script measures the effect with the restatement that reproduces the binary (oracle/snap_stream.c): the reference's
SBM-1024 graph, walk_len 80, num_walks 10, window 10, d = 16, several seeds, graph-reconstruction MAP with the binary's layout (rename) and
with node-id layout -- same walks, same stream otherwise.  CPU only.  Record: profiles/r03_unigram_layout_effect.json."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import snap_stream as ss                      # noqa: E402
from gem_amd.embedding.node2vec import node2vec          # noqa: E402
from gem_amd.evaluation import reconstruction as gr      # noqa: E402
from conftest import load_sbm1024                         # noqa: E402


def main(seeds=tuple(range(11, 27)), d=16):
    g = load_sbm1024()
    e = np.load(os.path.join(ROOT, 'tests', 'golden', 'sbm1024_edges.npy'))
    order, nbr, w = ss.load_edge_list(['%d %d %f' % (int(i), int(j), 1.0) for i, j in e.tolist()])
    out = {'seeds': list(seeds), 'd': d, 'binary_layout': [], 'node_id_layout': []}
    for seed in seeds:
        walks = ss.fast_walks(order, nbr, w, 1.0, 1.0, 10, 80, seed)
        for key, rename in (('binary_layout', True), ('node_id_layout', False)):
            ids, X = ss.fast_learn_embeddings(walks, d, 10, 1, seed, rename=rename)
            Y = np.zeros((g.number_of_nodes(), d))
            Y[np.asarray(ids)] = X
            m = node2vec(d=d, max_iter=1, walk_len=80, num_walks=10, con_size=10, ret_p=1, inout_p=1)
            out[key].append(float(gr.evaluateStaticGraphReconstruction(g, m, Y, None)[0]))
        print(seed, out['binary_layout'][-1], out['node_id_layout'][-1], flush=True)
    a, b = np.array(out['binary_layout']), np.array(out['node_id_layout'])
    out['mean_binary_layout'], out['mean_node_id_layout'] = float(a.mean()), float(b.mean())
    out['paired_relative_difference'] = float(((b - a) / a).mean())
    out['paired_relative_difference_se'] = float(((b - a) / a).std(ddof=1) / np.sqrt(len(a)))
    print(json.dumps(out))
    with open(os.path.join(ROOT, 'profiles', 'r03_unigram_layout_effect.json'), 'w') as fh:
        json.dump(out, fh)


if __name__ == '__main__':
    main()
