"""Source-node sharding of the GF and node2vec hot paths over N ranks (one process per GPU,
torch.distributed: backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

SURVEY 8(e):
  * GF edge-SGD shards by SOURCE ROW: rank r owns a contiguous block of source rows, nobody else
    writes them.  A sweep reads only the previous sweep's table when nodes are in ascending order
    (every rank then computes exactly what one GPU would), so the only exchange is an ALL-GATHER of
    the owned row blocks after each sweep.
  * node2vec walks shard by START NODE (a contiguous range of global walk ids); the graph is
    replicated; no collective while walking.  The vocabulary counts are summed once (all-reduce,
    4n bytes).  SGNS (`Node2VecPartitioned`): replicas that all-reduce the tables do NOT work beyond
    2 ranks -- summed deltas overshoot, averaged deltas under-train (measured) --
    so the tables are PARTITIONED instead (node v -> partition v % N, the scheme of GraphVite-style
    systems): the walk corpus is assembled on every rank once (all-gather of the shards) and trained episode
    by episode in N rounds in which rank g holds SynPos partition g and SynNeg partition (g+s) % N and runs
    TrainModel in walk order restricted to that bucket, the SynNeg partitions rotating around a ring between
    rounds.  No two GPUs ever touch the same row; with >= 64 episodes the embedding quality equals the
    sequential algorithm's.  (`Node2VecSharded`, the delta-sum replica scheme, is kept for N <= 2 and for the record.)

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

    def backend(self):
        return self.dist.get_backend() if self.world > 1 else 'none'

    def all_reduce_sum(self, t):
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)

    def all_gather_rows(self, full, own):
        """full[r*k:(r+1)*k] <- rank r's `own` (k rows each)."""
        if self.world > 1:
            self.dist.all_gather_into_tensor(full, own)
        else:
            full.copy_(own)

    def all_gather_ints(self, values, device, group=None):
        """Every rank's list of ints -> [world][len(values)] nested list (tiny control message)."""
        if self.world == 1:
            return [list(values)]
        import torch
        v = torch.as_tensor(list(values), dtype=torch.int64, device=device)
        out = torch.empty(self.world * v.numel(), dtype=torch.int64, device=device)
        self.dist.all_gather_into_tensor(out, v, group=group)
        return out.view(self.world, v.numel()).cpu().tolist()

    def all_to_all_rows(self, send, send_counts, recv_counts=None, cached=False, group=None):
        """send: [m, c] rows grouped by destination rank (send_counts[r] rows for rank r, in rank order).
        Returns the rows addressed to this rank, grouped by source rank."""
        if self.world == 1:
            return send
        import torch
        dist = self.dist
        rank = dist.get_rank()
        if recv_counts is None:
            recv_counts = [row[rank] for row in self.all_gather_ints(send_counts, send.device, group=group)]
        recv_counts = [int(x) for x in recv_counts]
        send_counts = [int(x) for x in send_counts]
        out = torch.empty((sum(recv_counts),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        if dist.get_backend() == 'nccl':
            dist.all_to_all_single(out, send.contiguous(), output_split_sizes=recv_counts, input_split_sizes=send_counts, group=group)
            return out
        ops, so, ro = [], 0, 0
        keep = []
        for r in range(self.world):
            a, b = send_counts[r], recv_counts[r]
            if r == rank:
                out[ro:ro + b].copy_(send[so:so + a])
            else:
                if a:
                    t = send[so:so + a].contiguous(); keep.append(t)
                    ops.append(dist.P2POp(dist.isend, t, r, group=group))
                if b:
                    ops.append(dist.P2POp(dist.irecv, out[ro:ro + b], r, group=group))
            so += a; ro += b
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return out

    def ring_shift(self, buf, tmp):
        """Every rank sends `buf` to rank-1 and receives rank+1's into `tmp`; returns (tmp, buf) swapped."""
        if self.world == 1:
            return buf, tmp
        dist = self.dist
        rank = dist.get_rank()
        ops = [dist.P2POp(dist.isend, buf, (rank - 1) % self.world), dist.P2POp(dist.irecv, tmp, (rank + 1) % self.world)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        return tmp, buf


# --------------------------------------------------------------------------- GF
class GFSharded(object):
    """GF sweeps with the source rows sharded over ranks.  With the edge arrays given, only the HALO crosses the fabric per
    sweep: the rows of other ranks that my firing edges (src owned, dst > src) read -- and because dst > src they all live on
    HIGHER ranks, so the exchange is one-directional.  On a graph with locality (SBM: 80 % of the edges inside a block) that
    is a fraction of the full-table all-gather; without locality (R-MAT) the plan falls back to the all-gather.  `gather()`
    assembles the full table once at the end."""

    def __init__(self, backend, comm, rank, world, n, src=None, dst=None, exchange_every=1):
        self.b, self.comm, self.rank, self.world = backend, comm, rank, world
        self.n = n
        # exchange_every = s > 1: the halo crosses the fabric only after every s-th sweep; in between a rank keeps training against the
        # last copy it received of the other ranks' rows (SURVEY 8e "or every s sweeps").  The result is then NOT the single-GPU one any
        # more: rows of other ranks are up to s-1 sweeps stale (block-Jacobi with delay across ranks, Gauss-Seidel inside one) -- the
        # difference is O(s * eta) per sweep relative to an update (tests/test_multi_gpu_cpu.py states it for the test graph).
        self.exchange_every = max(1, int(exchange_every))
        self._since = 0
        self.n_pad = (n + world - 1) // world * world
        self.block = self.n_pad // world
        self.r0 = rank * self.block
        self.r1 = min(self.r0 + self.block, n)
        self._edges = None if (src is None or world == 1) else (src, dst)
        if world > 1 and src is not None:
            # a rank's plan reads every row of ANOTHER rank from the previous sweep's table; that equals the single-GPU (and the
            # reference's) sweep only when no firing edge reads a row that the reference has already updated in the same sweep,
            # i.e. when sources are first visited in ascending id order (levels == 1 for the unsharded plan)
            import numpy as np
            s_, d_ = np.asarray(src), np.asarray(dst)
            fire = d_ > s_
            first = np.full(n, np.iinfo(np.int64).max, dtype=np.int64)
            np.minimum.at(first, s_[fire], np.nonzero(fire)[0])
            fs = first[np.unique(s_[fire])]
            if np.any(np.diff(fs) < 0):
                raise ValueError('GFSharded needs the firing sources in ascending id order (graph.edges() of a graph whose nodes were '
                                 'inserted in id order); other visiting orders make rows of other ranks "already updated" inside a sweep '
                                 'and would need a halo exchange per level')
        self.halo = None                                 # planned at the first sweep (needs the tables' device)
        self.halo_rows = None

    def _plan_halo(self, device):
        import numpy as np
        import torch
        src, dst = (np.asarray(a) for a in self._edges)
        fire = (src >= self.r0) & (src < self.r1) & (dst > src)
        need = np.unique(dst[fire])
        need = need[(need < self.r0) | (need >= self.r0 + self.block)].astype(np.int64)     # ascending == grouped by owner
        need_counts = np.bincount(need // self.block, minlength=self.world).tolist()
        cm = self.comm.all_gather_ints(need_counts, device)                                  # cm[asker][owner]
        asked_counts = [cm[a][self.rank] for a in range(self.world)]
        self.halo_rows = [sum(row) for row in cm]
        if max(self.halo_rows) > 0.75 * (self.world - 1) * self.block:   # no locality: the halo is most of what an all-gather moves
            self.halo = False
            return
        need_t = torch.from_numpy(need).to(device)
        asked = self.comm.all_to_all_rows(need_t.view(-1, 1), need_counts, asked_counts)     # rows of MINE the others read
        self.halo = (asked.view(-1), asked_counts, need_t, need_counts)

    def sweep(self, eta, regu):
        new = self.b.sweep(eta, regu)                   # tensor [n_pad, d] holding this sweep's table
        if self.world > 1:
            if self.halo is None and self._edges is not None:
                self._plan_halo(new.device)
            timed = bool(getattr(new, 'is_cuda', False))
            if timed:
                import torch
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
            self._since += 1
            if self.halo and self._since < self.exchange_every:
                # no exchange after this sweep: the halo rows of the table just written are whatever they were two sweeps ago -- carry the
                # last received copy over from the previous table (a device-local copy of the halo rows, no fabric traffic)
                need_idx = self.halo[2]
                old = self.b.X[self.b.cur ^ 1]
                new.index_copy_(0, need_idx, old.index_select(0, need_idx))
            elif self.halo:
                self._since = 0
                send_idx, send_counts, need_idx, need_counts = self.halo
                recv = self.comm.all_to_all_rows(new.index_select(0, send_idx), send_counts, need_counts, cached=True)
                new.index_copy_(0, need_idx, recv)
            else:
                self._since = 0
                self.gather(new)
            if timed:
                e1.record()
                self._comm_ev = getattr(self, '_comm_ev', [])
                self._comm_ev.append((e0, e1))
        return new

    def comm_seconds(self, reset=True):
        """Device seconds spent in the per-sweep exchange (halo all-to-all or all-gather) since the last reset."""
        evs = getattr(self, '_comm_ev', [])
        if not evs:
            return 0.0
        import torch
        torch.cuda.synchronize()
        t = sum(a.elapsed_time(b) for a, b in evs) * 1e-3
        if reset:
            self._comm_ev = []
        return t

    def gather(self, table):
        """All-gather of the owned row blocks: every rank ends with the full table (in place)."""
        if self.world > 1:
            own = table[self.r0:self.r0 + self.block].clone()
            self.comm.all_gather_rows(table, own)
        return table


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
        self.updates, self.rows, self.levels, self.algo_bytes, self.rows_per_wave = info[0], info[1], info[2], info[5], info[6]

    def sweep(self, eta, regu):
        s = self.torch.cuda.current_stream().cuda_stream
        _hip.check(self.L.gemhip_gf_plan_sweeps(self.plan, 1, eta, regu, C.c_void_p(s)))
        self.cur ^= 1
        return self.X[self.cur]

    def sweeps(self, k, eta, regu):
        """k sweeps in one library call: a plain launch loop (a captured hipGraph of 16 sweeps was measured SLOWER on ROCm 7.2 -- 10.5 vs
        8.7 us per sweep, DESIGN.md 3.1 -- and is not used); what the batch saves is the Python/ctypes overhead per sweep."""
        s = self.torch.cuda.current_stream().cuda_stream
        _hip.check(self.L.gemhip_gf_plan_sweeps(self.plan, k, eta, regu, C.c_void_p(s)))
        self.cur ^= (k & 1)
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
        self.lo, self.hi = shard_range(backend.num_start_nodes() * num_walks, rank, world)

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


def assemble_partitions(P_part, comm, world, n):
    """All-gather equal-size partition buffers and interleave them back into node-id order:
    row v = partition[v % world][v // world]."""
    import torch
    rows, d = P_part.shape
    full = torch.empty((world * rows, d), dtype=P_part.dtype, device=P_part.device)
    comm.all_gather_rows(full, P_part.contiguous())
    return full.view(world, rows, d).permute(1, 0, 2).reshape(-1, d)[:n].contiguous()


class Node2VecPartitioned(object):
    """One full node2vec.learn_embedding pass on N ranks with PARTITIONED tables (module docstring).

    Round 4: walks travel, pairs do not.  Every rank generates the walks of its start-node shard (as before), the vocabulary counts are summed
    (4n-byte all-reduce), and ONE all-gather assembles the walk corpus on every rank (4 bytes per token: 3.2 GB at SBM 1M/10M against the
    ~67 GB of (context, word) pairs the round-2/3 pipeline routed episode by episode).  An episode = the e-th slice of every rank's shard; in
    round s of it rank g runs TrainModel in WALK order over those walks restricted to the bucket (contexts of SynPos partition g, centre words of
    SynNeg partition (g+s) % N) -- `gemhip_sgns_train_part`, the single-GPU window kernel with a partition filter -- then the SynNeg partitions
    move one step around the ring.  No pair lists, no all-to-all, no host round trip inside the pass; the only per-round traffic is the ring
    shift of one SynNeg partition (n/N x d x 4 bytes)."""

    def __init__(self, backend, comm, rank, world, n, num_walks, walk_len, window, epochs, seed, flags, episodes=64, alpha0=0.025):
        self.b, self.comm, self.rank, self.world = backend, comm, rank, world
        self.n, self.num_walks, self.walk_len, self.window, self.epochs = n, num_walks, walk_len, window, epochs
        self.seed, self.flags, self.episodes, self.alpha0 = seed, flags, max(1, episodes), alpha0
        self.total_walks = backend.num_start_nodes() * num_walks
        self.lo, self.hi = shard_range(self.total_walks, rank, world)
        self.pairs_trained = 0

    def episode_table(self):
        """int64 [episodes][3][W]: for episode e and shard r -- first row of the slice in the gathered corpus, walks in the slice, global walk id of
        its first walk -- plus the work items per shard (the longest slice) of every episode."""
        W = self.world
        shard = [shard_range(self.total_walks, r, W) for r in range(W)]
        self.shard_rows = max(hi - lo for lo, hi in shard)             # rows per shard in the gathered corpus (shorter shards are padded)
        tab = np.zeros((self.episodes, 3, W), dtype=np.int64)
        seg_len = np.zeros(self.episodes, dtype=np.int64)
        for e in range(self.episodes):
            for r, (lo, hi) in enumerate(shard):
                a, z = shard_range(hi - lo, e, self.episodes)
                tab[e, 0, r] = r * self.shard_rows + a
                tab[e, 1, r] = z - a
                tab[e, 2, r] = lo + a
            seg_len[e] = max(1, int(tab[e, 1].max()))
        return tab, seg_len

    def run(self, p=1.0, q=1.0):
        b, comm, W, g = self.b, self.comm, self.world, self.rank
        b.walks(p, q, self.num_walks, self.walk_len, self.seed, self.flags, self.lo, self.hi)
        counts = b.vocab()
        comm.all_reduce_sum(counts)
        tab, seg_len = self.episode_table()
        corpus = b.gather_corpus(comm, self.shard_rows, W)                  # [W * shard_rows, walk_len] int32, identical on every rank
        # per-partition unigram tables: node-id order, or (flags & 16 = GEMHIP_N2V_VOCAB_ORDER, the plugin default on one GPU) the binary's layout --
        # each partition's nodes in order of first appearance in the gathered corpus
        b.build_unigram_parts(W, corpus if (self.flags & _hip.N2V_VOCAB_ORDER) else None, self.flags)
        if hasattr(b, 'locally_hot'):
            b.locally_hot(corpus)                                           # nodes whose tokens are packed into few walks: hot rows of the bucket launches
        seg_dev = b.upload_table(tab)
        P_part, N_cur, N_tmp = b.init_part_tables(self.seed, g, W)          # partition g of SynPos / SynNeg (+ a receive buffer)
        self.pairs_trained = 0
        self._ev = {'train': [], 'shift': []}
        timed = hasattr(b, 'event')
        # alpha: linear in the position of a token in the schedule (episode by episode, work-item order inside one), the same in every round of an
        # episode -- all ranks are at the same alpha at the same time
        tok_ep = [int(seg_len[e]) * W * self.walk_len for e in range(self.episodes)]
        alpha_total = self.epochs * sum(tok_ep)
        done = 0
        for ep in range(self.epochs):
            for e in range(self.episodes):
                for s in range(W):
                    j = (g + s) % W                                      # SynNeg partition visiting me this round
                    t0 = b.event() if timed else None
                    b.train_part(corpus, seg_dev, e, W, int(seg_len[e]), self.window, self.alpha0, alpha_total, done, ep, self.seed, self.flags,
                                 g, j, P_part, N_cur)
                    t1 = b.event() if timed else None
                    N_cur, N_tmp = comm.ring_shift(N_cur, N_tmp)         # after W shifts my own partition is back
                    if timed:
                        t2 = b.event()
                        self._ev['train'].append((t0, t1)); self._ev['shift'].append((t1, t2))
                done += tok_ep[e]
        self.pairs_trained = b.pairs(reset=True)
        return assemble_partitions(P_part, comm, W, self.n)

    def run_virtual(self, parts, p=1.0, q=1.0):
        """The N-rank schedule with N = `parts` VIRTUAL ranks on ONE GPU (world must be 1): the rounds of an episode run rank after rank instead of side by
        side -- the buckets of a round touch disjoint rows, so the result is what N GPUs compute (up to Hogwild's order inside a bucket).  Used by the
        GPU tests (quality of the partitioned schedule at N > 1 on the one-GPU test box) and by scripts/check_partitioned_1m.py for the per-rank cost
        model: `virtual_rank_seconds[g]` = kernel seconds rank g would spend."""
        assert self.world == 1, 'run_virtual emulates the ranks on one device'
        b, W = self.b, int(parts)
        b.walks(p, q, self.num_walks, self.walk_len, self.seed, self.flags, 0, self.total_walks)
        b.vocab()
        tab, seg_len = self.episode_table()                                 # one shard (this rank's = everything)
        corpus = b.gather_corpus(self.comm, self.shard_rows, 1)
        b.build_unigram_parts(W, corpus if (self.flags & _hip.N2V_VOCAB_ORDER) else None, self.flags)
        if hasattr(b, 'locally_hot'):
            b.locally_hot(corpus)
        seg_dev = b.upload_table(tab)
        tabs = [b.init_part_tables(self.seed, g, W) for g in range(W)]
        Pp, Np = [t[0] for t in tabs], [t[1] for t in tabs]
        tok_ep = [int(seg_len[e]) * self.walk_len for e in range(self.episodes)]
        alpha_total = self.epochs * sum(tok_ep)
        timed = hasattr(b, 'event')
        evs = [[] for _ in range(W)]
        done = 0
        for ep in range(self.epochs):
            for e in range(self.episodes):
                for s in range(W):
                    for g in range(W):
                        t0 = b.event() if timed else None
                        b.train_part(corpus, seg_dev, e, 1, int(seg_len[e]), self.window, self.alpha0, alpha_total, done, ep, self.seed, self.flags,
                                     g, (g + s) % W, Pp[g], Np[(g + s) % W])
                        if timed:
                            evs[g].append((t0, b.event()))
                done += tok_ep[e]
        self.pairs_trained = b.pairs(reset=True)
        if timed:
            import torch
            torch.cuda.synchronize()
            self.virtual_rank_seconds = [sum(a.elapsed_time(z) for a, z in ev) * 1e-3 for ev in evs]
        import torch
        rows = Pp[0].shape[0]
        full = torch.stack(Pp, dim=1).reshape(rows * W, -1)[:self.n].contiguous()       # row v = partition[v % W][v // W]
        return full

    def phase_seconds(self):
        """Device seconds of the last run() by phase (HIP events on the training stream): the bucket launches and the ring shifts of the SynNeg
        partitions between them."""
        if not getattr(self, '_ev', None) or not self._ev['train']:
            return None
        import torch
        torch.cuda.synchronize()
        return {k: sum(a.elapsed_time(b) for a, b in v) * 1e-3 for k, v in self._ev.items()}


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

    def num_start_nodes(self):
        m = C.c_int64()
        _hip.check(self.L.gemhip_n2v_start_nodes(self.h, C.byref(m)))
        return m.value

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    def walks(self, p, q, num_walks, walk_len, seed, flags, lo, hi):
        _hip.check(self.L.gemhip_n2v_walks(self.h, p, q, num_walks, walk_len, seed, flags, lo, hi, self._stream()))

    def vocab(self):
        _hip.check(self.L.gemhip_n2v_vocab(self.h, self._stream()))
        return self.counts

    def build_unigram(self):
        self.torch.cuda.current_stream().synchronize()
        if getattr(self, 'vocab_order', False):      # the binary's table layout (needs the WHOLE corpus on this handle: one rank)
            _hip.check(self.L.gemhip_n2v_build_unigram_vocab_order(self.h, getattr(self, 'vocab_flags', _hip.N2V_SNAP_COMPAT), None, None, None, None))
        else:
            _hip.check(self.L.gemhip_n2v_build_unigram(self.h, None, None, None))

    def init_tables(self, seed):
        self.torch.cuda.current_stream().synchronize()
        _hip.check(self.L.gemhip_sgns_init(self.h, self.d, seed, C.c_void_p(self.P.data_ptr()), C.c_void_p(self.N.data_ptr())))
        return self.P, self.N

    def train(self, window, epochs, epoch, lo, hi, tokens_total, token_offset, seed, flags):
        _hip.check(self.L.gemhip_sgns_train(self.h, window, 5, 0.025, epochs, epoch, lo, hi, tokens_total, token_offset, seed,
                                            flags, self._stream()))

    # ---- partitioned schedule -------------------------------------------------
    def build_unigram_parts(self, parts, corpus=None, flags=0):
        """corpus (the gathered walks of all ranks, a device tensor) given: the binary's table layout per partition (first-appearance order)."""
        self.torch.cuda.current_stream().synchronize()
        if corpus is not None:
            _hip.check(self.L.gemhip_n2v_build_unigram_parts_vocab_order(self.h, parts, flags, C.c_void_p(corpus.data_ptr()), corpus.numel(), None, None, None, None))
        else:
            _hip.check(self.L.gemhip_n2v_build_unigram_parts(self.h, parts, None, None))
        self.parts = parts

    def locally_hot(self, corpus):
        """gemhip_n2v_locally_hot_corpus on the gathered corpus: returns the number of locally hot nodes."""
        k = C.c_int64()
        _hip.check(self.L.gemhip_n2v_locally_hot_corpus(self.h, C.c_void_p(corpus.data_ptr()), corpus.shape[0], corpus.shape[1], -1, C.byref(k), self._stream()))
        return k.value

    def init_part_tables(self, seed, rank, world):
        """Partition `rank` of the SAME initial tables a single GPU would draw (InitPosEmb / InitNegEmb)."""
        torch = self.torch
        self.init_tables(seed)
        torch.cuda.current_stream().synchronize()
        rows = (self.n + world - 1) // world                       # equal-size buffers so ring shifts are uniform
        P = torch.zeros((rows, self.d), dtype=torch.float32, device=self.P.device)
        own = self.P[rank::world]
        P[:own.shape[0]].copy_(own)
        N = torch.zeros_like(P)
        return P, N, torch.zeros_like(P)

    def gather_corpus(self, comm, shard_rows, world):
        """My walk shard (padded to shard_rows with -1 tokens) -> the corpus of all ranks, rank-major, on every rank."""
        torch = self.torch
        nw = C.c_int64(); wl = C.c_int32(); p = C.c_void_p()
        _hip.check(self.L.gemhip_n2v_walks_ptr(self.h, C.byref(p), C.byref(nw), C.byref(wl)))
        mine = torch.full((shard_rows, wl.value), -1, dtype=torch.int32, device=self.P.device)
        _hip.check(self.L.gemhip_n2v_copy_walks(self.h, 0, nw.value, C.c_void_p(mine.data_ptr()), self._stream()))
        if world == 1:
            return mine
        full = torch.empty((world * shard_rows, wl.value), dtype=torch.int32, device=self.P.device)
        comm.all_gather_rows(full, mine)
        return full

    def upload_table(self, tab):
        return self.torch.from_numpy(np.ascontiguousarray(tab)).to(self.P.device)

    def event(self):
        e = self.torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def train_part(self, corpus, seg_dev, e, nseg, seg_len, window, alpha0, alpha_total, token_offset, epoch, seed, flags, ctx_part, word_part,
                   P_part, N_part):
        _hip.check(self.L.gemhip_sgns_train_part(self.h, C.c_void_p(corpus.data_ptr()), nseg * seg_len, corpus.shape[1],
                                                 C.c_void_p(seg_dev[e].data_ptr()), nseg, seg_len, 0, window, alpha0, alpha_total, token_offset,
                                                 epoch, seed, flags, ctx_part, word_part, C.c_void_p(P_part.data_ptr()),
                                                 C.c_void_p(N_part.data_ptr()), self.d, self._stream()))

    def pairs(self, reset=True):
        v = C.c_int64()
        _hip.check(self.L.gemhip_sgns_pairs(self.h, C.byref(v), 1 if reset else 0))
        return v.value

    def close(self):
        _hip.check(self.L.gemhip_n2v_destroy(self.h))
