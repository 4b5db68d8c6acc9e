"""node2vec on MI355X -- drop-in for gem.embedding.node2vec.node2vec
(gem/embedding/node2vec.py:8-57), which shells out to the SNAP binary
gem/c_exe/node2vec with `-i -o -d -l -r -k -e -p -q -v -dr -w`.

Here the same three phases (transition tables, biased walks, skip-gram with negative
sampling) run in libgem_hip.so: gem_amd/csrc/n2v.hip (n2v_alias_rows_kernel, n2v_walk_kernel, n2v_vocab_kernel) and
gem_amd/csrc/sgns.hpp (sgns_win_kernel) through gemhip_n2v_train (include/gem_hip.h).

Backend kwargs (kept on the instance like every GEM hyper-parameter): `seed`, `flags` (the binary's quirks and table layout, include/gem_hip.h),
and `n_gpus=N` with `devices=[...]` / `virtual_ranks=True` / `episodes=64` (gem_amd/embedding/_multi.py): walks sharded by start node, skip-gram over
partitioned tables -- gemhip_n2v_train_multi in one process, or gem_amd.multi_gpu.Node2VecPartitioned with one process per GPU under torch.distributed.
"""
import numpy as np

from gem_amd.embedding.static_graph_embedding import StaticGraphEmbedding


class node2vec(StaticGraphEmbedding):
    hyper_params = {
        'method_name': 'node2vec_rw',
    }

    def __init__(self, *args, **kwargs):
        super(node2vec, self).__init__(*args, **kwargs)

    def learn_embedding(self, graph=None, edge_f=None, is_weighted=False, no_python=False, **_ignored):
        if not graph:
            raise ValueError('graph needed')
        from gem_amd.embedding import _n2v_impl
        self._X = _n2v_impl.learn(self, graph)
        return self._X

    def get_edge_weight(self, i, j):
        return np.dot(self._X[i, :], self._X[j, :])
