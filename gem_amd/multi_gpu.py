"""Source-node sharding of the GF and node2vec hot paths over N ranks (one process per GPU,
torch.distributed: backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

SURVEY 8(e):
  * GF edge-SGD shards by SOURCE ROW: rank r owns a contiguous block of source rows, nobody else
    writes them.  A sweep reads only the previous sweep's table when nodes are in ascending order
    (every rank then computes exactly what one GPU would), so the only exchange is an ALL-GATHER of
    the owned row blocks after each sweep.
  * node2vec walks shard by START NODE (a contiguous range of global walk ids); the graph is
    replicated; no collective while walking.  The vocabulary counts are summed once (all-reduce,
    4n bytes).  SGNS trains each rank's walks against a local replica of SynPos/SynNeg and, every
    `sync_chunks`-th of its walks, all ranks exchange the SUM OF THEIR DELTAS since the last exchange
    (all-reduce on the d-dim tables): every rank's updates are applied once -- Hogwild with a bounded
    staleness of one chunk -- rather than averaged away.

The compute backend is injected: `HipBackend*` (below) drives libgem_hip.so through the C ABI with
torch tensors as device memory; the CPU tests inject a stand-in so the exchange logic runs under gloo.
PyTorch here is plumbing only: device buffers, streams, collectives.
"""
import ctypes as C

import numpy as np

from gem_amd import _hip


def shard_range(total, rank, world):
    """Contiguous, balanced split of range(total)."""
    return total * rank // world, total * (rank + 1) // world


class TorchComm(object):
    """Collectives used by the sharded drivers (sum all-reduce, equal-block all-gather)."""

    def __init__(self, world):
        self.world = world
        if world > 1:
            import torch.distributed as dist
            self.dist = dist

    def all_reduce_sum(self, t):
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)

    def all_gather_rows(self, full, own):
        """full[r*k:(r+1)*k] <- rank r's `own` (k rows each)."""
        if self.world > 1:
            self.dist.all_gather_into_tensor(full, own)


# --------------------------------------------------------------------------- GF
class GFSharded(object):
    def __init__(self, backend, comm, rank, world, n):
        self.b, self.comm, self.rank, self.world = backend, comm, rank, world
        self.n_pad = (n + world - 1) // world * world
        self.block = self.n_pad // world
        self.r0 = rank * self.block
        self.r1 = min(self.r0 + self.block, n)

    def sweep(self, eta, regu):
        new = self.b.sweep(eta, regu)                   # tensor [n_pad, d] holding this sweep's table
        if self.world > 1:
            own = new[self.r0:self.r0 + self.block].clone()
            self.comm.all_gather_rows(new, own)
        return new


class HipBackendGF(object):
    def __init__(self, n, src, dst, w, d, r0, r1, Xa, Xb):
        """Xa, Xb: torch float32 [n_pad, d] device tensors holding the SAME initial embedding."""
        import torch
        self.torch = torch
        self.L = _hip.lib()
        self.plan = C.c_void_p()
        _hip.check(self.L.gemhip_gf_plan_create(n, len(src), _hip.ptr(src, C.c_int32), _hip.ptr(dst, C.c_int32),
                                                _hip.ptr(_hip.as_f32(w), C.c_float), d, r0, max(r0, r1), C.byref(self.plan)))
        self.X = [Xa, Xb]
        self.cur = 0
        _hip.check(self.L.gemhip_gf_plan_bind(self.plan, C.c_void_p(Xa.data_ptr()), C.c_void_p(Xb.data_ptr())))
        info = (C.c_int64 * 8)()
        _hip.check(self.L.gemhip_gf_plan_info(self.plan, info))
        self.updates, self.rows, self.levels, self.algo_bytes = info[0], info[1], info[2], info[5]

    def sweep(self, eta, regu):
        s = self.torch.cuda.current_stream().cuda_stream
        _hip.check(self.L.gemhip_gf_plan_sweeps(self.plan, 1, eta, regu, C.c_void_p(s)))
        self.cur ^= 1
        return self.X[self.cur]

    def close(self):
        _hip.check(self.L.gemhip_gf_plan_destroy(self.plan))


# --------------------------------------------------------------------- node2vec
class Node2VecSharded(object):
    """One full node2vec.learn_embedding pass, sharded by start node."""

    def __init__(self, backend, comm, rank, world, n, num_walks, walk_len, window, epochs, seed, flags, sync_chunks=16):
        self.b, self.comm, self.rank, self.world = backend, comm, rank, world
        self.n, self.num_walks, self.walk_len, self.window, self.epochs = n, num_walks, walk_len, window, epochs
        self.seed, self.flags = seed, flags
        self.sync_chunks = max(1, sync_chunks) if world > 1 else 1
        self.lo, self.hi = shard_range(n * num_walks, rank, world)

    def run(self, p=1.0, q=1.0):
        b = self.b
        b.walks(p, q, self.num_walks, self.walk_len, self.seed, self.flags, self.lo, self.hi)
        counts = b.vocab()                              # device int32[n] (local)
        self.comm.all_reduce_sum(counts)                # -> global LearnVocab counts on every rank
        b.build_unigram()
        P, N = b.init_tables(self.seed)                 # identical on every rank (same seed)
        nloc = self.hi - self.lo
        tokens_local = max(nloc * self.walk_len, 1)
        if self.world > 1:
            P0, N0 = P.clone(), N.clone()
        for ep in range(self.epochs):
            for c in range(self.sync_chunks):
                a, z = shard_range(nloc, c, self.sync_chunks)
                # alpha decays with the LOCAL progress fraction: all ranks are at the same alpha at the same time
                b.train(self.window, self.epochs, ep, a, z, tokens_local, ep * tokens_local, self.seed, self.flags)
                if self.world > 1:
                    for T, T0 in ((P, P0), (N, N0)):
                        T.sub_(T0)                      # my delta since the last exchange
                        self.comm.all_reduce_sum(T)     # sum of everybody's deltas
                        T.add_(T0)
                        T0.copy_(T)
        return P


class HipBackendN2V(object):
    def __init__(self, n, row_ptr, col, w, d):
        import torch
        self.torch = torch
        self.L = _hip.lib()
        self.n, self.d = n, d
        self.h = C.c_void_p()
        _hip.check(self.L.gemhip_n2v_create(n, len(col), _hip.ptr(row_ptr, C.c_int64), _hip.ptr(col, C.c_int32),
                                            _hip.ptr(_hip.as_f32(w), C.c_float), C.byref(self.h)))
        dev = torch.device('cuda', torch.cuda.current_device())
        self.counts = torch.zeros(n, dtype=torch.int32, device=dev)
        _hip.check(self.L.gemhip_n2v_bind_counts(self.h, C.c_void_p(self.counts.data_ptr())))
        self.P = torch.empty((n, d), dtype=torch.float32, device=dev)
        self.N = torch.empty((n, d), dtype=torch.float32, device=dev)

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    def walks(self, p, q, num_walks, walk_len, seed, flags, lo, hi):
        _hip.check(self.L.gemhip_n2v_walks(self.h, p, q, num_walks, walk_len, seed, flags, lo, hi, self._stream()))

    def vocab(self):
        _hip.check(self.L.gemhip_n2v_vocab(self.h, self._stream()))
        return self.counts

    def build_unigram(self):
        self.torch.cuda.current_stream().synchronize()
        _hip.check(self.L.gemhip_n2v_build_unigram(self.h, None, None, None))

    def init_tables(self, seed):
        self.torch.cuda.current_stream().synchronize()
        _hip.check(self.L.gemhip_sgns_init(self.h, self.d, seed, C.c_void_p(self.P.data_ptr()), C.c_void_p(self.N.data_ptr())))
        return self.P, self.N

    def train(self, window, epochs, epoch, lo, hi, tokens_total, token_offset, seed, flags):
        _hip.check(self.L.gemhip_sgns_train(self.h, window, 5, 0.025, epochs, epoch, lo, hi, tokens_total, token_offset, seed,
                                            flags, self._stream()))

    def pairs(self, reset=True):
        v = C.c_int64()
        _hip.check(self.L.gemhip_sgns_pairs(self.h, C.byref(v), 1 if reset else 0))
        return v.value

    def close(self):
        _hip.check(self.L.gemhip_n2v_destroy(self.h))
