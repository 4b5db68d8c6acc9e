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
    systems): the (context, word) pairs of an episode of walks are materialised, routed to the owner of
    the context row (all-to-all) and trained in N rounds in which rank g holds SynPos partition g and
    SynNeg partition (g+s) % N, the SynNeg partitions rotating around a ring between rounds.  No two
    GPUs ever touch the same row; with >= 64 episodes the embedding quality equals the sequential
    algorithm's.  (`Node2VecSharded`, the delta-sum replica scheme, is kept for N <= 2 and for the record.)

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

    def side_group(self):
        """A second process group (its own RCCL communicator and stream) for collectives that run AHEAD of the training stream
        (Node2VecPartitioned._prepare): torch serialises all collectives of one group on one internal stream in issue order, so on
        the default group the next episode's all-to-all would queue behind this episode's ring shifts and lose the overlap."""
        if self.world == 1:
            return None
        if getattr(self, '_side', None) is None:
            self._side = self.dist.new_group(ranks=list(range(self.world)))
        return self._side

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
        self.updates, self.rows, self.levels, self.algo_bytes = info[0], info[1], info[2], info[5]

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
    """One full node2vec.learn_embedding pass on N ranks with PARTITIONED tables (module docstring)."""

    def __init__(self, backend, comm, rank, world, n, num_walks, walk_len, window, epochs, seed, flags, episodes=64, alpha0=0.025):
        self.b, self.comm, self.rank, self.world = backend, comm, rank, world
        self.n, self.num_walks, self.walk_len, self.window, self.epochs = n, num_walks, walk_len, window, epochs
        self.seed, self.flags, self.episodes, self.alpha0 = seed, flags, max(1, episodes), alpha0
        self.lo, self.hi = shard_range(backend.num_start_nodes() * num_walks, rank, world)
        self.pairs_trained = 0

    def _alpha(self, f):
        return self.alpha0 * max(1.0 - f, 1e-4)

    def _prepare(self, ep, e):
        """Everything of episode (ep, e) that does not touch the tables: materialise the (context, word) pairs of this slice of my
        walks grouped by (context % W, word % W) on the device, exchange the bucket sizes, route the pairs to the owners of
        their context rows (all-to-all).  Runs one episode AHEAD of the training rounds, on its own stream, so its host round
        trips (bucket sizes -> split sizes of the all-to-all) never stall the stream that trains."""
        b, comm, W, g = self.b, self.comm, self.world, self.rank
        nloc = self.hi - self.lo
        a, z = shard_range(nloc, e, self.episodes)
        pairs, counts = b.emit_pairs_bucketed(self.window, ep, a, z, self.seed, W)
        # The look-ahead collectives go through a SECOND process group (its own RCCL communicator and stream) under nccl: on the default
        # group torch serialises every collective on one internal stream in issue order, so the size exchange -- whose result the HOST
        # waits for -- would queue behind all W ring shifts of the episode that was just queued, and the host could not queue the next
        # episode before this one has drained.  Every rank issues the collectives of each group in the same order, which is what two
        # communicators need to make progress side by side.  GEM_N2V_SIDE_GROUP=0 falls back to the default group (gloo always does:
        # its collectives run on the host anyway).
        import os
        want = os.environ.get('GEM_N2V_SIDE_GROUP', '1' if getattr(comm, 'backend', lambda: 'gloo')() == 'nccl' else '0') == '1'
        side = comm.side_group() if (hasattr(comm, 'side_group') and want) else None
        kw = {'group': side} if side is not None else {}
        cm = comm.all_gather_ints(counts, pairs.device, **kw)        # cm[src][dest * W + wpart]
        send_counts = [sum(counts[r * W:(r + 1) * W]) for r in range(W)]
        recv_counts = [sum(cm[src][g * W:(g + 1) * W]) for src in range(W)]
        mine = comm.all_to_all_rows(pairs, send_counts, recv_counts, **kw)  # grouped by source rank, then by word % W
        seg, at = [[None] * W for _ in range(W)], 0
        for src in range(W):
            for j in range(W):
                ln = cm[src][g * W + j]
                seg[src][j] = (at, at + ln)
                at += ln
        return mine, seg

    def run(self, p=1.0, q=1.0):
        import torch
        b, comm, W, g = self.b, self.comm, self.world, self.rank
        b.walks(p, q, self.num_walks, self.walk_len, self.seed, self.flags, self.lo, self.hi)
        counts = b.vocab()
        comm.all_reduce_sum(counts)
        b.build_unigram_parts(W)
        P_part, N_cur, N_tmp = b.init_part_tables(self.seed, g, W)          # partition g of SynPos / SynNeg (+ a receive buffer)
        total_steps = float(self.epochs * self.episodes)
        self.pairs_trained = 0
        cuda = bool(getattr(P_part, 'is_cuda', False))
        main = torch.cuda.current_stream() if cuda else None
        prep = torch.cuda.Stream() if cuda else None
        ev = lambda: torch.cuda.Event(enable_timing=True) if cuda else None
        self._ev = {'train': [], 'shift': [], 'prep': []}

        def prepare(ep, e, first=False):
            if not cuda:
                return self._prepare(ep, e), None
            if first:
                prep.wait_stream(main)          # walks, vocabulary and tables are produced on `main`; nothing later on `main` feeds _prepare
            with torch.cuda.stream(prep):
                e0 = ev(); e0.record()
                out = self._prepare(ep, e)
                out[0].record_stream(main)
                e1 = ev(); e1.record()
            self._ev['prep'].append((e0, e1))
            return out, e1

        order = [(ep, e) for ep in range(self.epochs) for e in range(self.episodes)]
        nxt = prepare(*order[0], first=True)
        for k, (ep, e) in enumerate(order):
            (mine, seg), ready = nxt
            if ready is not None:
                main.wait_event(ready)
            step = ep * self.episodes + e
            for s in range(W):
                j = (g + s) % W                                          # SynNeg partition visiting me this round
                parts_j = [mine[x:y] for x, y in (seg[src][j] for src in range(W)) if y > x]
                bucket = parts_j[0] if len(parts_j) == 1 else (torch.cat(parts_j) if parts_j else mine[:0])
                f0, f1 = (step + s / W) / total_steps, (step + (s + 1) / W) / total_steps
                t0, t1, t2 = ev(), ev(), ev()
                if cuda: t0.record()
                b.train_pairs(bucket, j, P_part, N_cur, self._alpha(f0), self._alpha(f1), self.seed,
                              (step * W + g) * W + j, self.flags)
                if cuda: t1.record()
                self.pairs_trained += int(bucket.shape[0])
                N_cur, N_tmp = comm.ring_shift(N_cur, N_tmp)             # after W shifts my own partition is back
                if cuda:
                    t2.record()
                    self._ev['train'].append((t0, t1)); self._ev['shift'].append((t1, t2))
            # The rounds above are only QUEUED on `main` (kernel launches and P2P shifts are asynchronous).  The next episode's pairs are
            # emitted and routed now, on `prep`, which does NOT wait for `main` (round 2 made it wait for every round just queued, so the
            # host -- which synchronises `prep` inside _prepare to read the bucket sizes -- sat out the whole episode and `main` then idled
            # through the exchange: no overlap at all).  The host blocks here only for the emit kernels and the size exchange.
            nxt = prepare(*order[k + 1]) if k + 1 < len(order) else None
        return assemble_partitions(P_part, comm, W, self.n)

    def phase_seconds(self):
        """Device seconds of the last run() by phase (HIP events): training rounds, ring shifts of the SynNeg partitions (both on the
        training stream), pair emission + size exchange + all-to-all of the NEXT episode (side stream; issued without waiting for the
        training stream, so it runs next to the rounds -- how much of `prep` is hidden is `train + shift + prep - wall`)."""
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
        _hip.check(self.L.gemhip_n2v_build_unigram(self.h, None, None, None))

    def init_tables(self, seed):
        self.torch.cuda.current_stream().synchronize()
        _hip.check(self.L.gemhip_sgns_init(self.h, self.d, seed, C.c_void_p(self.P.data_ptr()), C.c_void_p(self.N.data_ptr())))
        return self.P, self.N

    def train(self, window, epochs, epoch, lo, hi, tokens_total, token_offset, seed, flags):
        _hip.check(self.L.gemhip_sgns_train(self.h, window, 5, 0.025, epochs, epoch, lo, hi, tokens_total, token_offset, seed,
                                            flags, self._stream()))

    # ---- partitioned schedule -------------------------------------------------
    def build_unigram_parts(self, parts):
        self.torch.cuda.current_stream().synchronize()
        _hip.check(self.L.gemhip_n2v_build_unigram_parts(self.h, parts, None, None))
        self.parts = parts

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

    def emit_pairs(self, window, epoch, lo, hi, seed):
        torch = self.torch
        nw = C.c_int64(); wl = C.c_int32(); p = C.c_void_p()
        _hip.check(self.L.gemhip_n2v_walks_ptr(self.h, C.byref(p), C.byref(nw), C.byref(wl)))
        cap = max((hi - lo) * wl.value * 2 * window, 1)
        buf = torch.empty((cap, 2), dtype=torch.int32, device=self.P.device)
        cnt = torch.zeros(1, dtype=torch.int64, device=self.P.device)
        _hip.check(self.L.gemhip_sgns_emit_pairs(self.h, window, epoch, lo, hi, seed, C.c_void_p(buf.data_ptr()), cap,
                                                 C.c_void_p(cnt.data_ptr()), self._stream()))
        return buf[:int(cnt.item())]

    def emit_pairs_bucketed(self, window, epoch, lo, hi, seed, parts):
        """Pairs grouped by key (context % parts) * parts + (word % parts); returns (int32 [np, 2] tensor, counts list)."""
        torch = self.torch
        nw = C.c_int64(); wl = C.c_int32(); p = C.c_void_p()
        _hip.check(self.L.gemhip_n2v_walks_ptr(self.h, C.byref(p), C.byref(nw), C.byref(wl)))
        cap = max((hi - lo) * wl.value * 2 * window, 1)
        buf = torch.empty((cap, 2), dtype=torch.int32, device=self.P.device)
        counts = (C.c_int64 * (parts * parts))()
        _hip.check(self.L.gemhip_sgns_emit_pairs_bucketed(self.h, window, epoch, lo, hi, seed, parts, C.c_void_p(buf.data_ptr()), cap, counts,
                                                          self._stream()))
        counts = list(counts)
        return buf[:sum(counts)], counts

    def train_pairs(self, bucket, neg_part, P_part, N_part, a0, a1, seed, stream_id, flags):
        if bucket.shape[0] == 0:
            return
        bucket = bucket.contiguous()
        _hip.check(self.L.gemhip_sgns_train_pairs(self.h, C.c_void_p(bucket.data_ptr()), bucket.shape[0], neg_part,
                                                  C.c_void_p(P_part.data_ptr()), C.c_void_p(N_part.data_ptr()), self.d, a0, a1, seed,
                                                  stream_id & 0xffffffff, flags, self._stream()))

    def pairs(self, reset=True):
        v = C.c_int64()
        _hip.check(self.L.gemhip_sgns_pairs(self.h, C.byref(v), 1 if reset else 0))
        return v.value

    def close(self):
        _hip.check(self.L.gemhip_n2v_destroy(self.h))
