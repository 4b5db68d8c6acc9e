"""Host side of node2vec.learn_embedding: nx graph -> CSR -> libgem_hip.so."""
import ctypes as C

import numpy as np

from gem_amd import _hip
from gem_amd.graph import edge_arrays, to_csr


def learn(model, graph):
    """Mirrors the argv of gem/embedding/node2vec.py:35-46:
    -d:_d -l:_walk_len -r:_num_walks -k:_con_size -e:_max_iter -p:_ret_p -q:_inout_p -dr -w."""
    import time
    t0 = time.perf_counter()
    n, src, dst, w, _ = edge_arrays(graph)
    row_ptr, col, ww = to_csr(n, src, dst, w)
    t1 = time.perf_counter()
    d = int(model._d)
    seed = getattr(model, '_seed', None)
    if seed is None:
        # the reference binary seeds with time(); draw from numpy's global RNG so np.random.seed() controls a run
        seed = int(np.random.randint(0, 2 ** 31 - 1))
    flags = int(getattr(model, '_flags', _hip.N2V_SNAP_LAYOUT))       # the binary's quirks AND its unigram-table layout (include/gem_hip.h)
    _hip.require_device()
    # n_gpus / devices / virtual_ranks / episodes kwargs (gem_amd/embedding/_multi.py): walks by start-node shard, SGNS over partitioned tables
    from gem_amd.embedding import _multi
    mode, n_gpus, devices = _multi.resolve(model)
    if mode != 'single':
        if mode == 'spmd':                        # every rank must train with the same seed: rank 0's draw is everyone's
            seed = int(_multi.broadcast_from_rank0(np.array([seed], dtype=np.int64))[0])
            X, st = _multi.n2v_spmd(model, n, row_ptr, col, ww, seed, flags)
        else:
            X, st = _multi.n2v_capi(model, n, row_ptr, col, ww, seed, flags, n_gpus, devices)
        t2 = time.perf_counter()
        model._stats = st
        model._node_num = n
        X64 = X.astype(np.float64)
        model._api_wall = _hip.api_wall(t0, t1, t2, time.perf_counter())
        return X64
    X = np.empty((n, d), dtype=np.float32)
    stats = (C.c_double * 4)()
    _hip.check(_hip.lib().gemhip_n2v_train(n, len(col), _hip.ptr(row_ptr, C.c_int64), _hip.ptr(col, C.c_int32),
                                           _hip.ptr(ww, C.c_float), d, int(model._walk_len), int(model._num_walks),
                                           int(model._con_size), int(model._max_iter), float(model._ret_p),
                                           float(model._inout_p), seed, flags, _hip.ptr(X, C.c_float), stats))
    t2 = time.perf_counter()
    model._stats = {'walk_seconds': stats[0], 'sgns_seconds': stats[1], 'tokens': stats[2]}
    model._node_num = n
    X64 = X.astype(np.float64)                    # node2vec.py:48 / loadEmbedding return float64
    model._api_wall = _hip.api_wall(t0, t1, t2, time.perf_counter())
    return X64
