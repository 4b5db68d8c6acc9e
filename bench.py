#!/usr/bin/env python3
"""bench.py -- headline benchmark of the gem_amd hot path (DESIGN.md "Measurement").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload node2vec|gf|hope]

One JSON line on stdout (rank 0).  A "step" is one pass of the hot path over a synthetic graph
that is already resident in HBM when the timed region starts:
    node2vec  one full node2vec.learn_embedding on SBM(1M nodes, 10M edges): walks (r=10, l=80,
              p=q=1) + vocabulary + unigram table + SGNS (k=10, 5 negatives, 1 epoch), d=128.
              BASELINE.json quotes its metric on this graph (configs[3]); edges/sec =
              graph.number_of_edges() / wall of the pass.
    gf        one SGD sweep over all edges of the same SBM, d=128; edges/sec = edges x sweeps / wall.
With no --workload the headline workload (node2vec) is timed first and the other two BASELINE configurations follow in the
same process -- GF on SBM 10k/100k with examples/run_sbm.py:66's eta/lambda (configs[1]) and on the 1M/10M graph, HOPE on SBM
100k/1M (configs[2]) -- each with its own timed region, roofline and cpu_baseline under "workloads" of the one JSON line.
For N>1 either launch one rank per GPU with torch.distributed.run (what the driver does) or run plain `python bench.py --gpus N`, which
spawns the N ranks itself; a mismatch between --gpus and WORLD_SIZE, or fewer GPUs than ranks, is an error (never a silent 1-GPU run).  Both paths shard by SOURCE /
START NODE (gem_amd/multi_gpu.py; node2vec additionally partitions its tables over the ranks and all-gathers the walk shards): total work is
fixed => "scaling": "strong".

`roofline`: for the dominant kernel, algorithmic bytes per launch (SURVEY 8d per-unit figure x units
the launch processes) / average launch duration measured with HIP events on the launch stream.
`cpu_baseline`: the reference CPU path timed on this box's host cores on a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from gem_amd import _hip, multi_gpu
from gem_amd.graph import sbm_graph, edge_arrays, to_csr

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable by a float4 copy)


def pmc_traffic(kernel, key):
    """HBM bytes per unit measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (profiles/*_pmc_traffic.json,
    corrected as MI355X_MICROARCH.md prescribes); bench.py scales it to the units one launch processes.  Returns (value, source):
    the figure is REPLAYED from the newest committed profile, not measured in this run."""
    best, src = None, None
    pdir = os.path.join(ROOT, 'profiles')
    for f in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if f.endswith('_pmc_traffic.json'):
            try:
                best = json.load(open(os.path.join(pdir, f)))[kernel][key]
                src = ('profiles/%s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same kernel, calibrated on launches of known byte counts in the '
                       'same access pattern, replayed per unit; not measured in this run)' % f)
            except (KeyError, ValueError):
                pass
    return best, src


def log(*a):
    print(*a, file=sys.stderr, flush=True)


_GRAPHS = {}
T_START = time.time()


def make_graph(args):
    key = (args.graph, args.nodes, args.edges, args.blocks)
    if key not in _GRAPHS:
        _GRAPHS.clear()                      # one graph at a time (R-MAT scale 22 holds 62M edges)
        _GRAPHS[key] = _make_graph(args)
    return _GRAPHS[key]


def _make_graph(args):
    t = time.time()
    if args.graph == 'rmat':                   # BASELINE configs[4] family: power-law R-MAT (a,b,c = .57,.19,.19)
        from gem_amd.graph import rmat_graph
        scale = int(np.ceil(np.log2(args.nodes)))
        g = rmat_graph(scale, args.edges, seed=20260923 + 5)
        log('R-MAT graph: scale %d, %d nodes, %d directed edges, max degree %d (%.1fs)' %
            (scale, g.n, g.number_of_edges(), int(np.bincount(g.src, minlength=g.n).max()), time.time() - t))
        return g
    g = sbm_graph(args.nodes, args.edges, args.blocks, seed=20260923 + 4)
    log('SBM graph: %d nodes, %d directed edges, %d blocks (%.1fs)' % (g.n, g.number_of_edges(), args.blocks, time.time() - t))
    return g


class GFWorkload(object):
    metric, unit, dtype, kernel = 'edges/sec', 'edges/s', 'f32', 'gf_sweep_kernel'
    default_steps, default_warmup = 50, 5

    def __init__(self, args, rank, world, comm):
        self.name = '%s%dk_%dk_gf_d%d_eta%g_regu%g' % (args.graph, args.nodes // 1000, args.edges // 1000, args.d, args.gf_eta, args.gf_regu)
        self.world, self.d, self.args = world, args.d, args
        # default: the "trainable" setting; --gf-eta 1e-4 --gf-regu 1.0 is examples/run_sbm.py:66's (same arithmetic per edge)
        self.eta, self.regu = args.gf_eta, args.gf_regu
        g = make_graph(args)
        self.n_edges = g.number_of_edges()
        n, src, dst, w, _ = edge_arrays(g)
        self.graph = (n, src, dst, w)
        dev = torch.device('cuda', torch.cuda.current_device())
        n_pad = (n + world - 1) // world * world
        gen = torch.Generator(device=dev); gen.manual_seed(1234)          # same init on every rank
        Xa = (0.01 * torch.randn(n_pad, self.d, device=dev, generator=gen, dtype=torch.float32)).contiguous()
        Xb = Xa.clone()
        r0, r1 = rank * (n_pad // world), min((rank + 1) * (n_pad // world), n)
        self.b = multi_gpu.HipBackendGF(n, src, dst, None, self.d, r0, r1, Xa, Xb)
        # N>1: halo exchange after EVERY sweep by default (bit-identical to one GPU: gf.py:93-100's Gauss-Seidel order); `--gf-exchange-every s`
        # (s > 1) is an opt-in that trades parity for speed -- other ranks' rows are then up to s-1 sweeps stale (block-Jacobi with delay,
        # SURVEY 8e "or every s sweeps") -- and `quality` then carries the deviation from the single-GPU result; one gather at the end
        self.exchange_every = args.gf_exchange_every if world > 1 else 1
        self.rank, self.sweeps_done, self.X_init = rank, 0, (Xa[:n].clone() if world > 1 else None)
        self.job = multi_gpu.GFSharded(self.b, comm, rank, world, n, src, dst, exchange_every=self.exchange_every)
        self.kernel_ms, self.launches = 0.0, 0
        self.kernel = 'gf_sweep_rows_kernel' if self.b.rows_per_wave > 1 else 'gf_sweep_kernel'
        log('[rank %d] GF plan: rows %d updates %d levels %d, %s (%d rows per wavefront)' % (rank, self.b.rows, self.b.updates, self.b.levels, self.kernel, self.b.rows_per_wave))

    def step(self):
        self.sweeps_done += 1
        if self.world > 1:
            self.last = self.job.sweep(self.eta, self.regu)
            return
        # one GPU: sweeps are handed to the library in batches of 64, the way GraphFactorization.learn_embedding hands it max_iter of them
        # (one library call = a plain launch loop: it saves the ctypes overhead per sweep, nothing else); every counted sweep runs inside the timed region
        self.pending = getattr(self, 'pending', 0) + 1
        if self.pending == 64:
            self._flush()

    def _flush(self):
        if getattr(self, 'pending', 0):
            self.last = self.b.sweeps(self.pending, self.eta, self.regu)
            self.pending = 0

    def finish(self):
        """End of a training run (inside the timed region): every rank assembles the full table."""
        if self.world == 1:
            self._flush()
        self.last = self.job.gather(self.last)

    def units_per_step(self):
        return self.n_edges

    def roofline(self, dev_ms_total, steps):
        launches = steps * self.b.levels
        avg_s = dev_ms_total * 1e-3 / launches
        algo = self.b.algo_bytes / self.b.levels        # 1548 B x updates
        compulsory = (self.b.rows * 2 * 4 * self.d + self.b.updates * (4 * self.d + 8)) / self.b.levels
        ach = compulsory / avg_s / 1e9
        per_upd, tsrc = pmc_traffic(self.kernel, 'traffic_bytes_per_update')
        traffic = None if per_upd is None else per_upd * self.b.updates / self.b.levels
        n = self.graph[0]
        table_mb = n * self.d * 4 / 1e6
        out = {'bound': 'hbm', 'kernel': self.kernel, 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
               'traffic': traffic, 'traffic_source': tsrc, 'achieved_traffic_GBs': None if traffic is None else traffic / avg_s / 1e9,
               'achieved_traffic_frac': None if traffic is None else traffic / avg_s / 1e9 / HBM_PEAK_GBS,
               'algorithmic_bytes_per_launch': compulsory, 'avg_launch_us': avg_s * 1e6,
               # SURVEY 8d's 1548 B per update charges X_i read + write to every edge; the kernel keeps X_i in registers for a whole row, so that figure
               # is NOT what a launch has to move (it gave "fractions" above 1 in rounds 1-3).  It stays as a rate comparable with the CPU loop's bytes.
               'comparability_bytes_per_launch': algo, 'comparability_GBs': algo / avg_s / 1e9,
               'note': '`achieved`/`frac` = compulsory bytes (every touched row read and written once + X_j and (j, w) read per update: rows x 8d + updates x '
                       '(4d + 8)) / launch time / 8 TB/s; comparability_GBs = SURVEY 8d\'s 1548 B per update x updates (X_i charged per edge: a reuse '
                       'factor of %.2f over the compulsory bytes, not a bandwidth)' % (algo / compulsory)}
        if table_mb * 2 <= 32.0:
            out['regime'] = ('launch-bound, not HBM-bound: both table copies (%.1f MB) stay in the 32 MB of L2 and a sweep is one %.1f us launch; the '
                             'HBM fraction above only says how far from an HBM-sized problem this configuration is' % (2 * table_mb, avg_s * 1e6))
        return out

    def cpu_baseline(self, budget_s=15.0):
        """(ii) the reference's own native path -- oracle/_ref/gf = g++ -O2 of gem/c_src/gf.cpp, run the way gf.py:55-72
        runs it (graph text file in, embedding text file out), end to end and loop-only (the run minus a 0-sweep run) -- is `value`
        (kind "reference"); beside it (i) GEM's Python loop gf.py:93-100 (restated in oracle/gf_pyloop.py: the reference file cannot
        travel to this box) and the C port used by the parity tests.  Graphs above 200k nodes are sampled (same density and block
        size): gf.cpp writes its embedding as text, 1.3 GB at 1M x 128."""
        import oracle
        from oracle import gf_pyloop
        from gem_amd.utils import graph_util
        n, src, dst, w = self.graph
        a = self.args
        X0 = (0.01 * np.random.RandomState(0).randn(n, self.d)).astype(np.float32)
        t = time.time(); oracle.gf_train_f32(n, src, dst, w, self.d, self.eta, self.regu, 1, X0); one = time.time() - t
        sweeps = max(1, min(20, int(0.3 * budget_s / max(one, 1e-3))))
        t = time.time(); oracle.gf_train_f32(n, src, dst, w, self.d, self.eta, self.regu, sweeps, X0); el = time.time() - t
        port = {'edges_per_s': self.n_edges * sweeps / el, 'sweeps': sweeps, 'kind': 'port',
                'what': 'oracle/gf_oracle.c (gf.cpp:152-164 restated), same %d-edge graph, 1 thread' % self.n_edges}
        # GEM's Python loop: a bounded number of edge visits of the first sweep
        _, visits, pel = gf_pyloop.gf_python_loop(src, dst, w, self.eta, self.regu, 3, X0.astype(np.float64), budget_s=0.25 * budget_s)
        pyl = {'edges_per_s': visits / pel, 'edge_visits_timed': visits, 'kind': 'port',
               'what': 'oracle/gf_pyloop.py = gf.py:93-100 operation by operation (fp64 numpy per edge), up to 3 sweeps or %.0f s' % (0.25 * budget_s)}
        out = {'value': port['edges_per_s'], 'unit': self.unit, 'cores': 1, 'kind': 'port', 'c_port': port, 'python_loop': pyl,
               'sample': '%d sweeps of the same %d-edge graph, oracle/gf_oracle.c (the reference loop is single-threaded)' % (sweeps, self.n_edges)}
        if os.path.exists(oracle.REF_GF):
            if n <= 200000:
                gs_n, gs_src, gs_dst, tag = n, src, dst, 'the same graph'
            else:
                gs = sbm_graph(200000, 200000 * (a.edges // a.nodes), max(1, 200000 // (a.nodes // a.blocks)), seed=7)
                gs_n, gs_src, gs_dst, _w, _ = edge_arrays(gs)
                tag = 'a 200k-node SBM of the same density and block size (gf.cpp saves its embedding as text)'
            tmp = tempfile.mkdtemp()
            gfile, efile = os.path.join(tmp, 'g.txt'), os.path.join(tmp, 'g.emb')
            with open(gfile, 'w') as fh:            # saveGraphToEdgeListTxt's format (graph_util.py:129-134): n, m, then "i j w" lines
                fh.write('%d\n%d\n' % (gs_n, len(gs_src)))
                np.savetxt(fh, np.stack([gs_src, gs_dst, np.ones(len(gs_src), np.int64)], axis=1), fmt='%d %d %d')

            def run(k):
                t = time.time()
                rc = subprocess.call([oracle.REF_GF, gfile, efile, '0', '1', str(self.d), repr(float(self.eta)), repr(float(self.regu)), str(k), '10000'],
                                     stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                return time.time() - t, rc
            t0, rc0 = run(0)
            per = max((run(1)[0] - t0), 1e-4)
            k = max(2, min(200, int(0.4 * budget_s / per)))
            tk, rck = run(k)
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)          # (the 200k-node embedding alone is ~260 MB of text)
            if rc0 == 0 and rck == 0 and tk > t0:
                loop = len(gs_src) * k / (tk - t0)
                out.update({'value': loop, 'kind': 'reference', 'cores': 1,
                            'reference_binary': {'loop_only_edges_per_s': loop, 'end_to_end_edges_per_s': len(gs_src) * k / tk, 'sweeps': k,
                                                 'seconds_end_to_end': tk, 'seconds_io_only': t0, 'nodes': gs_n, 'edges': int(len(gs_src))},
                            'sample': 'oracle/_ref/gf (g++ -O2 of gem/c_src/gf.cpp:130-169, argv of gf.py:55-72) on %s: %d sweeps in %.2f s end to end, '
                                      '%.2f s of it file IO (0-sweep run); value = loop only' % (tag, k, tk, t0)})
        return out

    def reset_counters(self):
        if self.world > 1:
            self.job.comm_seconds(reset=True)

    def phase_split(self, steps):
        comm = self.job.comm_seconds(reset=False)
        return {'exchange_seconds_per_sweep': comm / steps, 'exchange': 'halo all-to-all' if self.job.halo else 'all-gather',
                'exchange_every_sweeps': self.exchange_every, 'halo_rows_per_rank': self.job.halo_rows,
                'note': 'exchange_every_sweeps > 1: rows of other ranks are up to that many sweeps - 1 stale (not the single-GPU result; 1 is bit-identical)'}

    def check(self):
        assert bool(torch.isfinite(self.last).all()), 'non-finite embedding'

    def quality(self):
        """N>1 (outside the timed region, rank 0): the sharded table against the SAME number of sweeps on one GPU from the same initial table.
        exchange_every = 1 must reproduce it bit for bit; s > 1 reports what the stale halo cost."""
        if self.world == 1 or self.rank != 0:
            return None
        n, src, dst, w = self.graph
        Xa = torch.zeros_like(self.b.X[0]); Xa[:n].copy_(self.X_init); Xb = Xa.clone()
        ref = multi_gpu.HipBackendGF(n, src, dst, None, self.d, 0, n, Xa, Xb)
        R = ref.sweeps(self.sweeps_done, self.eta, self.regu)[:n]
        torch.cuda.synchronize()
        got = self.last[:n]
        dev = float((got - R).abs().max())
        moved = float((R - self.X_init).abs().max())
        ref.close()
        return {'sweeps': self.sweeps_done, 'exchange_every_sweeps': self.exchange_every, 'bit_identical_to_one_gpu': bool(torch.equal(got, R)),
                'max_abs_deviation_from_one_gpu': dev, 'largest_change_of_the_run': moved,
                'deviation_relative_to_largest_change': dev / moved if moved > 0 else 0.0}

    def api_wall(self):
        """`GraphFactorization.learn_embedding` end to end, numpy in / numpy out (gf.py:81-101 equivalent: edge arrays from the graph
        object, numpy's 0.01*randn table, H2D, max_iter sweeps, D2H, float64 copy) -- outside the timed region."""
        from gem_amd.embedding.gf import GraphFactorization
        n = self.graph[0]
        iters = 1000 if n <= 100000 else 100
        m = GraphFactorization(d=self.d, eta=self.eta, regu=self.regu, max_iter=iters, seed=5)
        m.learn_embedding(graph=make_graph(self.args), is_weighted=True, no_python=True)
        out = dict(m._api_wall)
        out.update({'max_iter': iters, 'edges_per_s_api': self.n_edges * iters / out['seconds'],
                    'what': 'gem_amd.embedding.gf.GraphFactorization(d=%d, max_iter=%d).learn_embedding(graph) wall, numpy in / float64 numpy out; '
                            'ingest_s includes numpy randn of the %d x %d initial table (gf.py:92)' % (self.d, iters, n, self.d)})
        return out


class N2VWorkload(object):
    metric, unit, dtype, kernel = 'edges/sec', 'edges/s', 'f32', 'sgns_win_kernel'
    default_steps, default_warmup = 2, 1

    def __init__(self, args, rank, world, comm):
        self.name = '%s%dk_%dk_node2vec_d%d_r%d_l%d_k%d' % (args.graph, args.nodes // 1000, args.edges // 1000, args.d, args.num_walks,
                                                            args.walk_len, args.window)
        if (args.ret_p, args.inout_q) != (1.0, 1.0):
            self.name += '_p%g_q%g' % (args.ret_p, args.inout_q)
        self.args, self.rank, self.world = args, rank, world
        g = make_graph(args)
        self.g = g
        self.n_edges = g.number_of_edges()
        n, src, dst, w, _ = edge_arrays(g)
        row_ptr, col, ww = to_csr(n, src, dst, w)
        self.b = multi_gpu.HipBackendN2V(n, row_ptr, col, ww, args.d)
        # the unigram alias table in the reference binary's own layout (first-appearance order, GEMHIP_N2V_VOCAB_ORDER) -- what node2vec.learn_embedding runs
        # by default: one table on one GPU, one per partition (each partition's nodes in first-appearance order of the gathered corpus) on N GPUs
        self.b.vocab_order = True
        if world == 1:
            self.job = multi_gpu.Node2VecSharded(self.b, comm, rank, world, n, args.num_walks, args.walk_len, args.window, 1,
                                                 seed=20260923, flags=_hip.N2V_SNAP_COMPAT)
        else:       # N GPUs: partitioned tables, episode schedule (gem_amd/multi_gpu.py)
            self.job = multi_gpu.Node2VecPartitioned(self.b, comm, rank, world, n, args.num_walks, args.walk_len, args.window, 1,
                                                     seed=20260923, flags=_hip.N2V_SNAP_LAYOUT, episodes=args.episodes)
        self.sgns_ms, self.sgns_launches, self.pairs = 0.0, 0, 0
        self._orig_train = self.b.train
        self.b.train = self._timed_train
        self.evs = []

    def _timed_train(self, *a):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); self._orig_train(*a); e1.record()
        self.evs.append((e0, e1))

    def reset_counters(self):
        torch.cuda.synchronize()
        self.evs = []
        self.b.pairs(reset=True)

    def step(self):
        self.P = self.job.run(float(self.args.ret_p), float(self.args.inout_q))

    def units_per_step(self):
        return self.n_edges

    def phase_split(self, steps):
        ph = self.job.phase_seconds() if hasattr(self.job, 'phase_seconds') else None
        if not ph:
            return None
        return {'last_step_seconds': ph, 'pairs_trained_by_this_rank': int(getattr(self.job, 'pairs_trained', 0)),
                'note': 'HIP events of the last pass on this rank: the bucket launches (TrainModel in walk order restricted to SynPos partition g x the visiting '
                        'SynNeg partition) and the ring shifts of the SynNeg partitions between them, serial on the training stream; the walk corpus is '
                        'assembled once per pass (one all-gather of the shards)'}

    def roofline(self, dev_ms_total, steps):
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self.evs)
        launches = len(self.evs)
        pairs = self.b.pairs(reset=False)
        avg_s = ms * 1e-3 / launches
        algo = (14 * 4 * self.args.d + 24) * pairs / launches       # 14*4d B per (centre,context) pair + ids
        ach = algo / avg_s / 1e9
        tokens = (self.job.hi - self.job.lo) * self.args.walk_len
        per_pair, tsrc = pmc_traffic(self.kernel + '_rmat', 'traffic_bytes_per_pair') if self.args.graph == 'rmat' else (None, None)     # (counters taken on R-MAT scale 22)
        if per_pair is None:
            per_pair, tsrc = pmc_traffic(self.kernel, 'traffic_bytes_per_pair')
        traffic = None if per_pair is None else per_pair * pairs / launches
        return {'bound': 'hbm', 'kernel': self.kernel, 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                'launch_plan': self.launch_plan(),
                'traffic': traffic, 'traffic_source': tsrc, 'achieved_traffic_GBs': None if traffic is None else traffic / avg_s / 1e9,
                'achieved_traffic_frac': None if traffic is None else traffic / avg_s / 1e9 / HBM_PEAK_GBS,
                'algorithmic_bytes_per_launch': algo, 'avg_launch_us': avg_s * 1e6,
                'pairs_per_launch': pairs / launches, 'tokens_per_launch': tokens,
                'sgns_fraction_of_step': ms / dev_ms_total,
                'note': 'algorithmic = 7168+24 B per (centre,context) pair at d=128 (SynPos r+w, 6 x SynNeg r+w); the kernel keeps the '
                        'positive SynNeg row in registers across a centre\'s contexts (12/14 of that reaches memory)'}

    def api_wall(self):
        """`node2vec.learn_embedding` end to end, numpy in / numpy out (node2vec.py:27-54 equivalent: graph object -> CSR, H2D,
        walks + vocabulary + unigram table + SGNS, D2H of SynPos, the float64 array loadEmbedding returns) -- one extra pass outside the timed region."""
        from gem_amd.embedding.node2vec import node2vec
        a = self.args
        m = node2vec(d=a.d, max_iter=1, walk_len=a.walk_len, num_walks=a.num_walks, con_size=a.window, ret_p=a.ret_p, inout_p=a.inout_q, seed=20260923)
        m.learn_embedding(graph=self.g, is_weighted=True, no_python=True)
        out = dict(m._api_wall)
        out.update({'edges_per_s_api': self.n_edges / out['seconds'],
                    'what': 'gem_amd.embedding.node2vec.node2vec(...).learn_embedding(graph) wall on the benchmark graph, numpy in / float64 numpy out '
                            '(the timed `value` starts with the CSR in HBM and leaves SynPos there)'})
        return out

    def launch_plan(self):
        """What gemhip_sgns_train actually launched in the last pass: read back from the handle (gemhip_sgns_last_launch -- the planner's choice after
        every knob and environment override, for the handle's own table layout), with the rule's inputs (n_eff over all / the cold rows)
        recomputed from the token counts for the SAME layout flags.  (Round 5 replayed the planner with the node-id layout's flags while the timed pass
        ran the vocabulary-order table: the line said 104 wavefronts for a pass that ran at 207.)"""
        try:
            if self.world != 1:
                return None
            L = _hip.lib()
            k, w, hot, fresh = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
            _hip.check(L.gemhip_sgns_last_launch(self.b.h, C.byref(k), C.byref(w), C.byref(hot), C.byref(fresh)))
            cnt = np.ascontiguousarray(self.b.counts.cpu().numpy(), dtype=np.int32)
            flags = _hip.N2V_SNAP_LAYOUT if getattr(self.b, 'vocab_order', False) else _hip.N2V_SNAP_COMPAT
            pk, pw, ph, ne, nec = C.c_int32(), C.c_int32(), C.c_int32(), C.c_double(), C.c_double()
            a = self.args
            _hip.check(L.gemhip_sgns_plan_launch(_hip.ptr(cnt, C.c_int32), cnt.size, a.d, a.window, a.walk_len, self.job.hi - self.job.lo,
                                                 flags, C.byref(pk), C.byref(pw), C.byref(ph), C.byref(ne), C.byref(nec)))
            return {'kernel': ['sgns_kernel', 'sgns_win_kernel (overwrite on leave)', 'sgns_win_kernel (Hogwild: delta write-back, reload-on-update)'][k.value],
                    'concurrent_wavefronts': w.value, 'hot_row_min_count': hot.value, 'hot_rows': int((cnt >= hot.value).sum()) if hot.value > 0 else 0,
                    'fresh_hot_rows_bits': fresh.value, 'unigram_layout_flags': flags,
                    'source': 'gemhip_sgns_last_launch (the launch as it ran)', 'planner_without_overrides': {'concurrent_wavefronts': pw.value, 'hot_row_min_count': ph.value},
                    'n_eff': ne.value, 'n_eff_cold': nec.value, 'rho': w.value * 5 * 0.4 / nec.value}
        except Exception as e:           # (an A/B library without the entry point)
            return {'error': str(e)[:200]}

    def cpu_baseline(self, budget_s=25.0):
        """The real reference binary (oracle/_ref/node2vec = gem/c_exe/node2vec) on a bounded sample: a 2048-node SBM of the same
        density and block size, same r/l/k/d, run twice -- on all host cores (how GEM runs it: racy Hogwild, its MAP collapses) and
        on ONE thread (race-free: the quality the reference is meant to have).  SGNS cost is linear in tokens, so edges/s carries
        over to the full graph (tokens/edge identical).  `value` is the all-cores rate; both rates and both MAPs are reported."""
        import oracle
        from gem_amd.embedding.node2vec import node2vec
        from gem_amd.evaluation import reconstruction as gr
        a = self.args
        n_s = 2048
        gs = sbm_graph(n_s, n_s * (a.edges // a.nodes), max(1, n_s // (a.nodes // a.blocks)), seed=7)
        cores = min(os.cpu_count() or 1, 16)    # SNAP's dynamic OpenMP loop gets SLOWER beyond a few threads on small graphs
        model = node2vec(d=a.d, max_iter=1, walk_len=a.walk_len, num_walks=a.num_walks, con_size=a.window, ret_p=1, inout_p=1)
        if os.path.exists(oracle.REF_N2V):
            tmp = tempfile.mkdtemp()
            gf = os.path.join(tmp, 'g.graph')
            t = time.time()
            with open(gf, 'w') as fh:            # saveGraphToEdgeListTxtn2v (graph_util.py:137-140): "%d %d %f" per edge
                fh.writelines('%d %d %f\n' % (i, j, 1.0) for i, j in zip(gs.src.tolist(), gs.dst.tolist()))
            t_write = time.time() - t
            runs = {}
            for thr in (cores, 1):
                emb = os.path.join(tmp, 'g%d.emb' % thr)
                t = time.time()
                subprocess.call([oracle.REF_N2V, '-i:' + gf, '-o:' + emb, '-d:%d' % a.d, '-l:%d' % a.walk_len,
                                 '-r:%d' % a.num_walks, '-k:%d' % a.window, '-e:1', '-p:1.000000', '-q:1.000000', '-dr', '-w'],
                                stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, OMP_NUM_THREADS=str(thr)))
                el = time.time() - t
                t = time.time()
                X = np.zeros((n_s, a.d))
                with open(emb) as fh:            # loadEmbedding (graph_util.py:161-169)
                    fh.readline()
                    for line in fh:
                        tok = line.split()
                        X[int(tok[0])] = [float(v) for v in tok[1:]]
                t_load = time.time() - t
                runs[thr] = {'edges_per_s': gs.number_of_edges() / el, 'seconds': el, 'threads': thr,
                             'MAP': gr.evaluateStaticGraphReconstruction(gs, model, X, None)[0],
                             # node2vec.py:27-54 as a whole: edge-list text dump + the binary (its own text IO inside) + loadEmbedding
                             'api_wall': {'seconds': t_write + el + t_load, 'graph_text_write_s': t_write, 'binary_s': el, 'load_embedding_s': t_load,
                                          'edges_per_s_api': gs.number_of_edges() / (t_write + el + t_load)}}
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)
            # the HIP path on the very same sample graph (outside every timed region): the MAP the baselines are to be compared with
            mh = node2vec(d=a.d, max_iter=1, walk_len=a.walk_len, num_walks=a.num_walks, con_size=a.window, ret_p=1, inout_p=1, seed=20260923)
            Xh = mh.learn_embedding(graph=gs, is_weighted=True, no_python=True)
            return {'value': runs[cores]['edges_per_s'], 'unit': self.unit, 'cores': cores, 'kind': 'reference',
                    'all_cores': runs[cores], 'single_thread_race_free': runs[1], 'committed_full_size_runs': self._full_size_runs(),
                    'full_size_reference': self._full_size_reference(a.nodes),
                    'hip_map_same_sample': gr.evaluateStaticGraphReconstruction(gs, mh, Xh, None)[0],
                    'sample': 'gem/c_exe/node2vec (SNAP ELF) end to end incl. its text IO on an SBM with %d nodes / %d edges (same '
                              'density, block size, d, r, l, k): %d threads %.1fs, 1 thread %.1fs; MAP = graph reconstruction over all nodes of '
                              'the sample' % (n_s, gs.number_of_edges(), cores, runs[cores]['seconds'], runs[1]['seconds'])}
        n, src, dst, w, _ = edge_arrays(gs)
        t = time.time()
        oracle.n2v_train(n, src, dst, None, a.d, a.walk_len, 1, a.window, 1, 1.0, 1.0, 1, 11)
        el = (time.time() - t) * a.num_walks
        return {'value': gs.number_of_edges() / el, 'unit': self.unit, 'cores': 1, 'kind': 'port',
                'sample': 'oracle/n2v_oracle.c, r=1 timed and scaled x%d (linear in tokens), SBM %d nodes' % (a.num_walks, n_s)}

    @staticmethod
    def _full_size_reference(nodes):
        """The reference binary's OWN runs on the benchmark graph itself (committed records of hours-long CPU runs, tests/golden/n2v_ref_snap_<n>k*.json):
        {threads: {edges_per_s, seconds, MAP}} -- carried in the compact line next to the 2048-node sample's rate."""
        out = {}
        for thr, f in ((1, 'n2v_ref_snap_%dk.json' % (nodes // 1000)), (4, 'n2v_ref_snap_%dk_t4.json' % (nodes // 1000))):
            p = os.path.join(ROOT, 'tests', 'golden', f)
            if os.path.exists(p):
                j = json.load(open(p))
                out['threads_%d' % thr] = {'edges_per_s': round(j['edges_per_s'], 1), 'seconds': round(j['seconds']), 'MAP': round(j['MAP'], 4)}
        return out or None

    @staticmethod
    def _full_size_runs():
        """The 2048-node sample above is cache-resident, i.e. the EASY case for the CPU.  The reference runs made for the parity goldens
        (scripts/make_golden_n2v_scale.py, hours of CPU each, not repeatable inside a bench run) are quoted from their committed records."""
        out = []
        gdir = os.path.join(ROOT, 'tests', 'golden')
        for f in sorted(os.listdir(gdir)):
            if f.startswith('n2v_ref_') and f.endswith('.json') and ('snap_1' in f or 'oracle_1' in f):       # the 100k / 1000k runs (race-free, 4-thread, oracle)
                try:
                    j = json.load(open(os.path.join(gdir, f)))
                    out.append({'file': 'tests/golden/' + f, 'engine': j['engine'], 'nodes': j['params']['n'], 'edges_per_s': j['edges_per_s'],
                                'seconds': j['seconds'], 'MAP': j['MAP']})
                except (KeyError, ValueError):
                    pass
        return out

    def check(self):
        assert bool(torch.isfinite(self.P).all()), 'non-finite embedding'
        assert float(self.P.abs().max()) > 1e-3

    @staticmethod
    def _ref_golden(engine, n, vocab_order=False):
        """Committed reference run of `engine` ('snap' | 'oracle') on the n-node benchmark graph: the one scored over the larger node sample; for the
        oracle, the run in the unigram-table layout the timed pass used (vocab_order: flags 27, the binary's; else flags 11) when it exists."""
        gdir = os.path.join(ROOT, 'tests', 'golden')
        names = ['n2v_ref_%s_%dk_s4096.json' % (engine, n // 1000), 'n2v_ref_%s_%dk.json' % (engine, n // 1000)]
        if engine == 'oracle' and vocab_order:
            names.insert(0, 'n2v_ref_oracle_%dk_vocab_order_s4096.json' % (n // 1000))
        for name in names:
            if os.path.exists(os.path.join(gdir, name)):
                return os.path.join(gdir, name)
        return os.path.join(gdir, names[-1])

    def quality(self, nsample=1024):
        """Outside the timed region: graph-reconstruction MAP of the learned table over a FIXED node sample, with the reference
        evaluator's semantics (gem_amd/csrc/eval.hip), next to the MAP the reference binary reaches on the very same graph and
        sample (tests/golden/n2v_ref_snap_<n>k.json, made by scripts/make_golden_n2v_scale.py: gem/c_exe/node2vec race-free)."""
        from gem_amd.evaluation import reconstruction as gr
        a = self.args
        vo = bool(getattr(self.b, 'vocab_order', False))
        for engine in ('snap', 'oracle'):             # score the sample the committed reference runs were scored on
            path = self._ref_golden(engine, self.g.n, vo)
            if a.graph == 'sbm' and os.path.exists(path):
                nsample = max(nsample, len(json.load(open(path))['ap']))
        rng = np.random.RandomState(0)
        nodes = rng.choice(self.g.n, size=min(nsample, self.g.n), replace=False)        # uniform over all nodes, hubs included (a prefix of a larger sample)
        ap = gr.sampled_ap_gpu(self.g, None, self.P.cpu().numpy(), nodes)
        out = {'sampled_map': float(ap.mean()), 'sampled_map_se': float(ap.std(ddof=1) / np.sqrt(len(ap))), 'nodes_sampled': int(len(nodes)),
               'evaluator': 'metrics.computeMAP semantics on the GPU', 'reference_map': None,
               'unigram_layout': 'vocabulary order (the binary\'s: GEMHIP_N2V_VOCAB_ORDER)' if vo else 'node-id order'}
        ap_by_engine = {'snap': ap, 'oracle': ap}
        opath = self._ref_golden('oracle', self.g.n, vo)
        oflags = json.load(open(opath))['params'].get('flags', 11) if (a.graph == 'sbm' and os.path.exists(opath)) else None
        if vo and oflags is not None and not (oflags & 16) and self.world == 1:
            # only a node-id-layout oracle run is committed for this size: one more pass (outside every timed region) in THAT layout, same seed, pairs
            # with it draw for draw.  (At the headline size the oracle's flags-27 run is committed since round 5 and the timed pass itself pairs with it.)
            self.b.vocab_order = False
            P2 = self.job.run(float(a.ret_p), float(a.inout_q))
            self.b.vocab_order = True
            ap_by_engine['oracle'] = gr.sampled_ap_gpu(self.g, None, P2.cpu().numpy(), nodes)
            out['oracle_pass'] = {'unigram_layout': 'node-id order (the layout of the oracle\'s committed run: paired draws)',
                                  'sampled_map': float(ap_by_engine['oracle'].mean())}
        for engine, key in (('snap', 'reference_map'), ('oracle', 'oracle_map')):
            ap = ap_by_engine[engine]
            path = self._ref_golden(engine, self.g.n, vo)
            if a.graph == 'sbm' and os.path.exists(path):
                ref = json.load(open(path))
                pr = ref['params']
                if (pr['n'], pr['edges'], pr['blocks'], pr['seed'], pr['d'], pr['walk_len'], pr['num_walks'], pr['window']) == \
                        (a.nodes, a.edges, a.blocks, 20260923 + 4, a.d, a.walk_len, a.num_walks, a.window) and len(ref['ap']) <= len(ap):
                    m = len(ref['ap'])            # RandomState(0).choice(n, m) is a prefix of choice(n, m') for m' > m (permutation prefix)
                    out[key] = ref['MAP']
                    out[key + '_se'] = ref['MAP_se']
                    out[key + '_source'] = 'tests/golden/%s: %s, same graph, same %d-node sample' % (os.path.basename(path), ref['engine'], m)
                    d = ap[:m] - np.asarray(ref['ap'])
                    out['map_minus_' + key] = float(d.mean())
                    out['map_minus_' + key + '_se'] = float(d.std(ddof=1) / np.sqrt(len(d)))
        if a.graph == 'rmat' and self.world == 1:
            # power-law family (BASELINE configs[4]): the sequential oracle's run on the same R-MAT graph, scored over the nodes that have a ranked neighbour
            # (reconstruction.eligible_sample: a uniform sample is two thirds dead weight there, tests/test_rmat_gpu.py) -- committed for scale 17 and 20;
            # the reference binary itself cannot run these graphs (its per-pair alias tables are Sigma deg^2)
            scale = int(np.ceil(np.log2(a.nodes)))
            gpath = None
            for suffix in ('e128k', 'e16k'):            # (scale 20 is scored over 131 072 eligible nodes, scale 17 over 16 384)
                cand = os.path.join(ROOT, 'tests', 'golden', 'n2v_ref_oracle_rmat%d%s_%s.json' % (scale, '_vocab_order' if vo else '', suffix))
                if gpath is None and os.path.exists(cand):
                    gpath = cand
            if gpath is not None:
                ref = json.load(open(gpath))
                pr = ref['params']
                if (pr['rmat_scale'], pr['edges'], pr['seed'], pr['d'], pr['walk_len'], pr['num_walks'], pr['window']) == \
                        (scale, a.edges, 20260923 + 5, a.d, a.walk_len, a.num_walks, a.window):
                    en = gr.eligible_sample(self.g, len(ref['ap']))
                    ape = gr.sampled_ap_gpu(self.g, None, self.P.cpu().numpy(), en)
                    dd = ape - np.asarray(ref['ap'])
                    out.update({'oracle_map': ref['MAP'], 'oracle_map_se': ref['MAP_se'], 'eligible_map': float(ape.mean()), 'eligible_nodes_sampled': int(len(en)),
                                'map_minus_oracle_map': float(dd.mean()), 'map_minus_oracle_map_se': float(dd.std(ddof=1) / np.sqrt(len(dd))),
                                'oracle_map_source': 'tests/golden/%s: %s, same graph and seed, paired over %d nodes that have a ranked neighbour' %
                                                     (os.path.basename(gpath), ref['engine'], len(en))})
        if out['reference_map'] is None:
            out['reference_map_note'] = ('no committed reference run for this graph size; the largest one is tests/golden/n2v_ref_snap_100k.json '
                                         '(python bench.py --nodes 100000 --edges 1000000 --blocks 10 reports against it)')
        return out


class HopeWorkload(object):
    """BASELINE configs[2]: SBM 100k nodes / 1M edges, HOPE d=128 (k=64), beta=0.01; embeddings/sec = n / wall."""
    metric, unit, dtype, kernel = 'embeddings/sec', 'embeddings/s', 'f32', 'hope_spmm16_kernel'
    default_steps, default_warmup = 5, 1

    def __init__(self, args, rank, world, comm):
        if args.nodes == 1000000 and args.edges == 10000000:
            args.nodes, args.edges, args.blocks = 100000, 1000000, 32
        self.directed = bool(getattr(args, 'hope_directed', False))
        self.name = 'sbm%dk_%dk_hope_d%d_beta0.01%s' % (args.nodes // 1000, args.edges // 1000, args.d, '_directed' if self.directed else '')
        self.args = args
        g = make_graph(args)
        if self.directed:        # hope.py:28-36 assumes no symmetry: the same SBM with every undirected edge kept in ONE random direction (A != A^T)
            from gem_amd.graph import orient_randomly
            g = orient_randomly(g, 1)
        self.n_edges = g.number_of_edges()
        n, src, dst, w, _ = edge_arrays(g)
        self.n = n
        self.row_ptr, self.col, _ = to_csr(n, src, dst, None)
        self.k = args.d // 2
        self.U = np.empty((n, self.k), np.float32); self.V = np.empty((n, self.k), np.float32); self.sig = np.empty(self.k, np.float32)
        # the timed step leaves U sqrt(S), V sqrt(S) in HBM (gemhip_hope_plan_solve_device: the bench contract's "resident" rate); the numpy-out
        # form GEM's API needs -- the same solve plus two 4nk-byte copies into pageable host memory -- is timed after it and reported beside it
        self.dU = torch.empty((n, self.k), dtype=torch.float32, device='cuda'); self.dV = torch.empty_like(self.dU)
        self.stats = (C.c_double * 12)()
        self.plan = C.c_void_p()                     # graph resident in HBM before the timed region (staged API)
        _hip.check(_hip.lib().gemhip_hope_plan_create(n, len(self.col), _hip.ptr(self.row_ptr, C.c_int64), _hip.ptr(self.col, C.c_int32), None,
                                                      0.01, C.byref(self.plan)))
        self.dev_s, self.spmm, self.spmm_cols, self.calls = 0.0, 0.0, 0.0, 0
        self.world = world                      # HOPE does not shard (replicas only): N ranks = N independent replicas

    def reset_counters(self):
        self.dev_s, self.spmm, self.spmm_cols, self.calls = 0.0, 0.0, 0.0, 0
        self.spmm_s, self.eig_s = 0.0, 0.0

    def step(self):
        _hip.check(_hip.lib().gemhip_hope_plan_solve_device(self.plan, self.k, 16, 3, 20, 1e-5, 20260923, C.c_void_p(self.dU.data_ptr()),
                                                            C.c_void_p(self.dV.data_ptr()), _hip.ptr(self.sig, C.c_float), self.stats))
        self.dev_s += self.stats[0]; self.spmm += self.stats[1]; self.spmm_cols += self.stats[2]; self.calls += 1
        self.spmm_s = getattr(self, 'spmm_s', 0.0) + self.stats[11]; self.eig_s = getattr(self, 'eig_s', 0.0) + self.stats[8]

    def units_per_step(self):
        return self.n * self.world

    def host_output_seconds(self, reps=3):
        """The numpy-out form (gemhip_hope_plan_solve: what HOPE.learn_embedding calls), wall seconds per solve, outside the timed region."""
        st = (C.c_double * 12)()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize(); t = time.time()
            _hip.check(_hip.lib().gemhip_hope_plan_solve(self.plan, self.k, 16, 3, 20, 1e-5, 20260923, _hip.ptr(self.U, C.c_float),
                                                         _hip.ptr(self.V, C.c_float), _hip.ptr(self.sig, C.c_float), st))
            ts.append(time.time() - t)
        return float(np.median(ts))

    def roofline(self, dev_ms_total, steps):
        # SpMM compulsory bytes = 8 nnz + 4(n+1) + 2*4*n*b per launch (b = dense block columns of that launch)
        host_s = self.host_output_seconds()
        same = bool(np.array_equal(self.U, self.dU.cpu().numpy()) and np.array_equal(self.V, self.dV.cpu().numpy()))      # same solve, same bits
        launches = self.spmm
        bavg = self.spmm_cols / launches
        algo = 8.0 * self.n_edges + 4.0 * (self.n + 1) + 8.0 * self.n * bavg
        avg_s = self.spmm_s / launches
        ach = algo / avg_s / 1e9
        gather = (4.0 * bavg + 8.0) * self.n_edges + 8.0 * self.n * bavg
        per_col, tsrc = pmc_traffic(self.kernel, 'traffic_bytes_per_column')
        traffic = None if per_col is None else per_col * bavg
        return {'bound': 'hbm', 'kernel': self.kernel, 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                'traffic': traffic, 'traffic_source': tsrc, 'achieved_traffic_GBs': None if traffic is None else traffic / avg_s / 1e9,
                'achieved_traffic_frac': None if traffic is None else traffic / avg_s / 1e9 / HBM_PEAK_GBS,
                'algorithmic_bytes_per_launch': algo, 'avg_launch_us': avg_s * 1e6, 'spmm_launches_per_step': launches / self.calls,
                'avg_block_columns': bavg, 'device_seconds_per_step': self.dev_s / self.calls, 'spmm_seconds_per_step': self.spmm_s / self.calls,
                'host_eig_seconds_per_step': self.eig_s / self.calls, 'restarts': self.stats[5], 'katz_terms': self.stats[3],
                'pcie_inclusive': {'seconds_per_step': host_s, 'embeddings_per_s': self.n / host_s, 'outputs_identical_to_device_form': same,
                                   'note': 'gemhip_hope_plan_solve: the same solve with both n x k outputs copied into pageable numpy buffers (GEM API form); '
                                           '`value` is the device-resident form gemhip_hope_plan_solve_device'},
                'solver': 'symmetric_chebyshev_filter' if self.stats[3] == 0 else 'block_krylov',
                'note': 'algorithmic = SURVEY 8d compulsory bytes of one SpMM (8 nnz + 4(n+1) + 8 n b: the dense block is read once); the row '
                        'gathers themselves move %.3g B per launch = %.0f GB/s out of L2 / Infinity Cache (the %d MB block fits on chip); launch '
                        'time = HIP events around the back-to-back SpMM runs inside the solver' % (gather, gather / avg_s / 1e9, int(4 * self.n * bavg / 1e6))}

    def cpu_baseline(self, budget_s=30.0):
        """hope.py:28-36 cannot form its dense S at n=100k (three 80 GB matrices), and scipy svds with sparse-LU solves
        takes minutes already at n=10k (the LU of I - beta A fills in).  Baseline = the same SVD through scipy's
        ARPACK svds on the Katz-series operator (oracle/hope_oracle.py hope_operator_series), on a 20k-node SBM of the
        same density and block size, all host cores (BLAS/ARPACK threads)."""
        from oracle import hope_oracle
        import scipy.sparse as sp
        a = self.args
        n_s = min(20000, self.n)
        gs = sbm_graph(n_s, n_s * (a.edges // a.nodes), max(1, n_s // (a.nodes // a.blocks)), seed=7)
        if self.directed:
            from gem_amd.graph import orient_randomly
            gs = orient_randomly(gs, 1)
        A = sp.csr_matrix((np.ones(gs.number_of_edges()), (gs.src, gs.dst)), shape=(n_s, n_s))
        t = time.time()
        _, s_cpu = hope_oracle.hope_operator_series(A, 0.01, a.d, tol=1e-5)
        el = time.time() - t
        # the same sample through the HIP path: sigma-relative error against the CPU svds
        from gem_amd.embedding.hope import HOPE
        m = HOPE(d=a.d, beta=0.01)
        m.learn_embedding(graph=gs, is_weighted=True, no_python=True)
        rel = float(np.abs(np.asarray(m._sigma, dtype=np.float64) / s_cpu - 1.0).max())
        return {'value': n_s / el, 'unit': self.unit, 'cores': os.cpu_count() or 1, 'kind': 'port',
                'sigma_rel_err_vs_cpu_svds': rel,
                'sample': 'scipy ARPACK svds(k=%d, tol=1e-5) on the implicit Katz operator, SBM %d nodes / %d edges (same density and '
                          'block size), %.1fs; ARPACK matvec count grows with n, so this rate is optimistic for n=%d; the HIP path on '
                          'the same sample agrees to max |sigma/sigma_cpu - 1| = %.1e'
                          % (a.d // 2, n_s, gs.number_of_edges(), el, self.n, rel)}

    def api_wall(self):
        """`HOPE.learn_embedding` end to end, numpy in / numpy out (hope.py:23-41 equivalent), outside the timed region."""
        from gem_amd.embedding.hope import HOPE
        g = make_graph(self.args)
        if self.directed:
            from gem_amd.graph import orient_randomly
            g = orient_randomly(g, 1)
        m = HOPE(d=self.args.d, beta=0.01)
        m.learn_embedding(graph=g, is_weighted=True, no_python=True)
        out = dict(m._api_wall)
        out.update({'embeddings_per_s_api': self.n / out['seconds'],
                    'what': 'gem_amd.embedding.hope.HOPE(d=%d, beta=0.01).learn_embedding(graph) wall, numpy in / float64 numpy out; host_prepare_s = the plan '
                            '(transpose, symmetry test, uploads, spectral-radius estimate), kernels_s = the solve' % self.args.d})
        return out

    def check(self):
        assert bool(torch.isfinite(self.dU).all()) and np.all(np.diff(self.sig) >= 0) and self.sig[0] > 0


WORKLOADS = {'gf': GFWorkload, 'node2vec': N2VWorkload, 'hope': HopeWorkload}


def capi_multi(args, rank, world):
    """`--driver capi`: the N-GPU entry points of the C ABI themselves (include/gem_hip.h gemhip_gf_train_multi / gemhip_n2v_train_multi: ONE host
    process drives n_gpus devices, RCCL loaded by the library) -- the interface INTEGRATION.md hands a GEM maintainer -- instead of the
    torch.distributed stand-in of gem_amd/multi_gpu.py.  Under torch.distributed.run the call is made by rank 0 with devices 0..N-1 while the other
    ranks wait at the barrier; `--virtual-ranks` (or a box with fewer GPUs than ranks under the gloo test backend) repeats device 0 in the list: the
    library then runs the ranks as VIRTUAL ranks on one GPU (collectives become copies; sharding, schedule and kernels are the production code).
    The entry points are one-shot drop-ins (host arrays in and out), so the timed region is the one the library reports in `stats`: GF = the K sweeps
    including every exchange; node2vec = walks + vocabulary + corpus gather + training of one pass.  Returns the full bench record (rank 0) or None."""
    name = 'node2vec' if args.workload == 'all' else args.workload
    if name not in ('gf', 'node2vec'):
        raise SystemExit('bench.py --driver capi: gf and node2vec shard (SURVEY 8e); hope is replicas only')
    if rank != 0:
        return None
    g = make_graph(args)
    n, src, dst, w, _ = edge_arrays(g)
    N = args.gpus
    virt = args.virtual_ranks or torch.cuda.device_count() < N
    devs = (C.c_int32 * N)(*([0] * N if virt else list(range(N))))
    L = _hip.lib()
    K = args.steps if args.steps is not None else WORKLOADS[name].default_steps
    W = args.warmup if args.warmup is not None else WORKLOADS[name].default_warmup
    out = {'metric': 'edges/sec', 'unit': 'edges/s', 'n_gpus': N, 'steps': K, 'warmup': W, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
           'dtype': 'f32', 'data': 'synthetic'}
    t_all = time.perf_counter()
    if name == 'gf':
        X = (0.01 * np.random.RandomState(1234).randn(n, args.d)).astype(np.float32)
        st = (C.c_double * 8)()
        call = lambda it: _hip.check(L.gemhip_gf_train_multi(n, len(src), _hip.ptr(src, C.c_int32), _hip.ptr(dst, C.c_int32), None, args.d, args.gf_eta, args.gf_regu,
                                                             it, N, devs, _hip.ptr(X, C.c_float), st))
        if W:
            call(W)
        call(K)
        el = st[0]
        assert np.isfinite(X).all()
        out.update({'value': g.number_of_edges() * K / el, 'ms_per_step': el * 1e3 / K,
                    'config': {'workload': '%s%dk_%dk_gf_d%d_eta%g_regu%g' % (args.graph, args.nodes // 1000, args.edges // 1000, args.d, args.gf_eta, args.gf_regu),
                               'nodes': n, 'directed_edges': g.number_of_edges(), 'd': args.d, 'sharding': 'source-node x%d' % N,
                               'driver': 'capi gemhip_gf_train_multi' + (' (virtual ranks on one GPU)' if virt else '')},
                    'phases': {'exchange_bytes_per_rank_per_sweep': st[3], 'updates_per_sweep': st[1], 'virtual_ranks': st[5]}})
    else:
        row_ptr, col, ww = to_csr(n, src, dst, w)
        X = np.empty((n, args.d), np.float32)
        st = (C.c_double * 8)()
        secs = []
        for it in range(W + K):
            _hip.check(L.gemhip_n2v_train_multi(n, len(col), _hip.ptr(row_ptr, C.c_int64), _hip.ptr(col, C.c_int32), None, args.d, args.walk_len, args.num_walks,
                                                args.window, 1, float(args.ret_p), float(args.inout_q), 20260923, _hip.N2V_SNAP_LAYOUT, N, devs, args.episodes,
                                                _hip.ptr(X, C.c_float), st))
            if it >= W:
                secs.append(st[0] + st[1])
        el = float(np.sum(secs))
        assert np.isfinite(X).all()
        from gem_amd.evaluation import reconstruction as gr
        nodes = np.random.RandomState(0).choice(n, size=min(1024, n), replace=False)
        ap = gr.sampled_ap_gpu(g, None, X, nodes)
        out.update({'value': g.number_of_edges() * K / el, 'ms_per_step': el * 1e3 / K,
                    'config': {'workload': '%s%dk_%dk_node2vec_d%d_r%d_l%d_k%d' % (args.graph, args.nodes // 1000, args.edges // 1000, args.d, args.num_walks, args.walk_len, args.window),
                               'nodes': n, 'directed_edges': g.number_of_edges(), 'd': args.d, 'sharding': 'start-node x%d, partitioned tables' % N,
                               'driver': 'capi gemhip_n2v_train_multi' + (' (virtual ranks on one GPU)' if virt else '')},
                    'phases': {'walks_vocab_gather_s': st[0], 'train_s': st[1], 'pairs_trained': st[3], 'ring_shift_bytes_per_rank_per_round': st[4],
                               'bucket_launches_per_rank': st[7], 'virtual_ranks': st[6]},
                    'quality': {'sampled_map': float(ap.mean()), 'nodes_sampled': int(len(nodes)),
                                'unigram_layout': 'vocabulary order per partition (GEMHIP_N2V_VOCAB_ORDER: the plugin default)'}})
    out['config']['world_size_seen'] = world
    out['call_wall_s_incl_uploads_and_planning'] = time.perf_counter() - t_all
    return out


def time_workload(name, args, rank, world, comm, K=None, W=None, with_cpu=True):
    """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides; max over ranks.
    Returns (result dict on rank 0 else None, workload)."""
    wl = WORKLOADS[name](args, rank, world, comm)
    K = K if K is not None else wl.default_steps
    W = W if W is not None else wl.default_warmup

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(W):
        wl.step()
    if W and hasattr(wl, 'finish'):
        wl.finish()
    barrier()
    if hasattr(wl, 'reset_counters'):
        wl.reset_counters()
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(K):
        wl.step()
    if hasattr(wl, 'finish'):
        wl.finish()
    ev1.record()
    barrier()
    el = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    wl.check()
    t = torch.tensor([el], dtype=torch.float64, device='cuda')
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    if rank != 0:
        return None, wl
    out = {
        'metric': wl.metric, 'value': wl.units_per_step() * K / el, 'unit': wl.unit, 'n_gpus': world, 'steps': K, 'warmup': W,
        'ms_per_step': el * 1e3 / K, 'higher_is_better': True,
        'scaling': 'weak' if (world == 1 or name == 'hope') else 'strong',
        'vs_baseline': None, 'dtype': wl.dtype, 'data': 'synthetic',
        'config': {'workload': wl.name, 'nodes': args.nodes, 'directed_edges': wl.n_edges, 'd': args.d,
                   'sharding': 'source-node x%d' % world},
    }
    if world > 1:
        out['config']['world_size_seen'] = dist.get_world_size()
        if hasattr(wl, 'phase_split'):
            out['phases'] = wl.phase_split(K)
    if world == 1:
        out['roofline'] = wl.roofline(dev_ms, K)          # (before quality(): that may run one more, untimed pass on the same handle)
    if hasattr(wl, 'quality'):
        q = wl.quality()
        if q is not None:
            out['quality'] = q
    if world == 1:
        if hasattr(wl, 'api_wall') and not getattr(args, 'no_api_wall', False):
            out['api_wall'] = wl.api_wall()
        if with_cpu:
            out['cpu_baseline'] = wl.cpu_baseline()
    return out, wl


ROOFLINE_KEYS = ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'achieved_traffic_GBs', 'achieved_traffic_frac', 'avg_launch_us',
                 'algorithmic_bytes_per_launch', 'regime_short')
CPU_KEYS = ('value', 'unit', 'cores', 'kind', 'sample', 'full_size_reference', 'full_size')
QUALITY_KEYS = ('sampled_map', 'nodes_sampled', 'unigram_layout_short', 'reference_map', 'map_minus_reference_map', 'map_minus_reference_map_se', 'oracle_map',
                'map_minus_oracle_map', 'map_minus_oracle_map_se', 'bit_identical_to_one_gpu', 'deviation_relative_to_largest_change')
MAX_LINE_BYTES = 4000        # the driver parses the LAST stdout line from a bounded tail: round 4's 23 KB line came back as parsed = null


def _pick(d, keys, maxlen=160):
    out = {}
    for k in keys:
        if d is not None and k in d:
            v = d[k]
            out[k] = (v[:maxlen - 3] + '...') if isinstance(v, str) and len(v) > maxlen else v
    return out


def compact_line(full, detail_path=None):
    """The ONE stdout line of the bench contract, <= MAX_LINE_BYTES: the contract fields of the headline workload, its `roofline` and `cpu_baseline`
    objects reduced to their numbers, the parity gaps of `quality`, and a five-number summary of every other workload timed in the same process.
    Everything else (notes, sources, API walls, launch plans, the full per-workload objects) goes to the detail record (stderr + file)."""
    line = {k: full[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                                 'dtype', 'data') if k in full}
    line['config'] = _pick(full.get('config'), ('workload', 'nodes', 'directed_edges', 'd', 'sharding', 'driver', 'world_size_seen'))

    def roof(r):
        if r is None:
            return None
        r = dict(r)
        if 'regime' in r:
            r['regime_short'] = 'launch-bound (tables L2-resident)'
        return _pick(r, ROOFLINE_KEYS)
    if 'roofline' in full:
        line['roofline'] = roof(full['roofline'])
    if 'cpu_baseline' in full:
        line['cpu_baseline'] = _pick(full['cpu_baseline'], CPU_KEYS, 200)
    if full.get('quality'):
        q = dict(full['quality'])
        if 'unigram_layout' in q:
            q['unigram_layout_short'] = 'vocab-order' if q['unigram_layout'].startswith('vocab') else 'node-id'
        line['quality'] = _pick(q, QUALITY_KEYS)
    if full.get('phases'):
        line['phases'] = {k: v for k, v in full['phases'].items() if isinstance(v, (int, float))}
    if full.get('workloads'):
        wl = {}
        for name, w in full['workloads'].items():
            if not isinstance(w, dict):
                wl[name] = str(w)[:80]
                continue
            e = {'value': w.get('value'), 'unit': w.get('unit'), 'ms_per_step': w.get('ms_per_step')}
            r = w.get('roofline') or {}
            e.update({'kernel': r.get('kernel'), 'frac': r.get('frac'), 'traffic_frac': r.get('achieved_traffic_frac')})
            c = w.get('cpu_baseline') or {}
            if c.get('value'):
                e['cpu'] = c['value']
            q = w.get('quality') or {}
            for k in ('oracle_map', 'map_minus_oracle_map', 'sampled_map'):
                if q.get(k) is not None:
                    e[k] = q[k]
            wl[name] = {k: (float('%.6g' % v) if isinstance(v, float) else v) for k, v in e.items() if v is not None}
        line['workloads'] = wl
    if detail_path:
        line['detail'] = detail_path

    def rnd(o):
        if isinstance(o, float):
            return float('%.10g' % o)
        if isinstance(o, dict):
            return {k: rnd(v) for k, v in o.items()}
        return o
    line = rnd(line)
    s = json.dumps(line)
    for victim in ('workloads', 'phases', 'quality'):          # never exceed the budget: drop the optional summaries, most voluminous first
        if len(s) <= MAX_LINE_BYTES:
            break
        line.pop(victim, None)
        s = json.dumps(line)
    assert len(s) <= MAX_LINE_BYTES, len(s)
    return s


def emit(full):
    """Rank 0: the detail record (everything measured, tens of KB) to a file and to stderr; then the compact line -- the only line on stdout."""
    detail_path = None
    text = json.dumps(full)
    for d in (os.environ.get('GEM_BENCH_DETAIL_DIR'), os.path.join(ROOT, 'gpurun_out'), tempfile.gettempdir()):
        if not d:
            continue
        try:
            os.makedirs(d, exist_ok=True)
            detail_path = os.path.join(d, 'bench_detail_latest.json')
            with open(detail_path, 'w') as fh:
                fh.write(text + '\n')
            break
        except OSError:
            detail_path = None
    log('BENCH_DETAIL ' + text)
    rel = os.path.relpath(detail_path, ROOT) if detail_path and detail_path.startswith(ROOT) else detail_path
    print(compact_line(full, rel), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--workload', default=os.environ.get('GEM_BENCH_WORKLOAD', 'all'), choices=sorted(WORKLOADS) + ['all'],
                    help="'all' (default) = node2vec (the headline line) followed by gf and hope under \"workloads\"")
    ap.add_argument('--nodes', type=int, default=1000000)
    ap.add_argument('--edges', type=int, default=10000000)
    ap.add_argument('--blocks', type=int, default=100)
    ap.add_argument('--graph', default='sbm', choices=['sbm', 'rmat'])
    ap.add_argument('--d', type=int, default=128)
    ap.add_argument('--num-walks', type=int, default=10)
    ap.add_argument('--walk-len', type=int, default=80)
    ap.add_argument('--window', type=int, default=10)
    ap.add_argument('--ret-p', type=float, default=1.0, help='node2vec return parameter p (node2vec.py:40 -p:)')
    ap.add_argument('--inout-q', type=float, default=1.0, help='node2vec in-out parameter q (node2vec.py:41 -q:)')
    ap.add_argument('--gf-eta', type=float, default=1e-2)
    ap.add_argument('--gf-regu', type=float, default=1e-2)
    ap.add_argument('--gf-exchange-every', type=int, default=1, help='N>1 GF: sweeps between halo exchanges (default 1 = after every sweep: bit-identical to one GPU; '
                    's > 1 is NOT the single-GPU algorithm: rows of other ranks are up to s-1 sweeps stale, `quality` reports the deviation)')
    ap.add_argument('--hope-directed', action='store_true', help='hope: orient every undirected edge in one random direction (A != A^T: the general case of hope.py)')
    ap.add_argument('--episodes', type=int, default=64, help='N>1 node2vec: episodes of the partitioned schedule')
    ap.add_argument('--driver', default='torch', choices=['torch', 'capi'], help="N>1: 'torch' = one process per GPU over torch.distributed (gem_amd/multi_gpu.py, "
                    "what the driver's command launches); 'capi' = rank 0 calls the library's own N-GPU entry points (gemhip_*_train_multi: one process, n_gpus devices)")
    ap.add_argument('--virtual-ranks', action='store_true', help='--driver capi: run the N ranks as virtual ranks on device 0 (the one-GPU test box)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-api-wall', action='store_true', help='skip the extra learn_embedding() pass that fills `api_wall` (outside the timed region)')
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)')
    if args.gpus > 1 and args.driver == 'capi' and 'WORLD_SIZE' not in os.environ:
        # one process IS the product's N-GPU shape: no ranks to spawn
        if torch.cuda.device_count() < args.gpus and not args.virtual_ranks:
            raise SystemExit('bench.py --gpus %d --driver capi: only %d GPU(s) visible (--virtual-ranks runs them on device 0)' % (args.gpus, torch.cuda.device_count()))
        torch.cuda.set_device(0)
        _hip.check(_hip.lib().gemhip_set_device(0))
        emit(capi_multi(args, 0, 1))
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- N ranks of this very command under torch.distributed.run, one per GPU
        # (what the driver's N>1 command does explicitly).  Never fall back to one GPU silently: fewer devices than ranks is an error
        # unless the gloo test backend was asked for (tests/test_bench_gpu.py runs 2 ranks on the one GPU of the test box).
        if torch.cuda.device_count() < args.gpus and os.environ.get('GEM_BENCH_BACKEND', 'nccl') == 'nccl':
            raise SystemExit('bench.py --gpus %d: only %d GPU(s) visible' % (args.gpus, torch.cuda.device_count()))
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log('bench.py: spawning %d ranks: %s' % (args.gpus, ' '.join(cmd)))
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))))
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d -- launch one rank per GPU (python -m torch.distributed.run --nproc-per-node %d '
                         'bench.py --gpus %d), or run plain `python bench.py --gpus %d` and let it spawn the ranks' % (args.gpus, world, args.gpus, args.gpus, args.gpus))
    local = local % torch.cuda.device_count()      # (lets the N>1 code path be exercised on a 1-GPU box with GEM_BENCH_BACKEND=gloo)
    torch.cuda.set_device(local)
    _hip.check(_hip.lib().gemhip_set_device(local))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('GEM_BENCH_BACKEND', 'nccl')          # "nccl" is RCCL on ROCm
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if args.driver == 'capi' and args.gpus > 1:
        if world > 1:
            dist.barrier()
        out = capi_multi(args, rank, world)
        if rank == 0:
            emit(out)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    comm = multi_gpu.TorchComm(world)
    headline = 'node2vec' if args.workload == 'all' else args.workload
    out, wl = time_workload(headline, args, rank, world, comm, args.steps, args.warmup, with_cpu=not args.no_cpu_baseline)

    if args.workload == 'all' and world == 1 and (args.nodes, args.edges, args.graph) == (1000000, 10000000, 'sbm'):
        # the other two BASELINE configurations, each with its own timed region (their step counts are their own defaults)
        del wl
        torch.cuda.empty_cache()
        import copy
        extra = {}
        a2 = copy.copy(args); a2.nodes, a2.edges, a2.blocks, a2.gf_eta, a2.gf_regu = 10000, 100000, 10, 1e-4, 1.0
        extra['gf_sbm10k_100k_run_sbm_setting'], w2 = time_workload('gf', a2, rank, world, comm, 1000, 100, with_cpu=not args.no_cpu_baseline)
        del w2
        extra['gf_sbm1m_10m'], w3 = time_workload('gf', copy.copy(args), rank, world, comm, None, None, with_cpu=not args.no_cpu_baseline)
        del w3
        torch.cuda.empty_cache()
        extra['hope_sbm100k_1m'], w4 = time_workload('hope', copy.copy(args), rank, world, comm, None, None, with_cpu=not args.no_cpu_baseline)
        del w4
        a5 = copy.copy(args); a5.hope_directed = True      # the general (directed) Katz case: block-Krylov SVD on S^T S, no eigen-path
        extra['hope_sbm100k_directed'], w5 = time_workload('hope', a5, rank, world, comm, None, None, with_cpu=not args.no_cpu_baseline)
        del w5
        torch.cuda.empty_cache()
        # BASELINE configs[4] on ONE GPU: R-MAT scale 22 (4.2M nodes, ~62M directed edges after symmetrisation), one node2vec pass and GF sweeps.
        # No CPU baseline: SNAP's per-(t, v) alias tables are Sigma deg^2 on a graph with 94k-degree hubs -- the reference runs out of host
        # memory there -- and gf.cpp's rate does not depend on the graph (c_port of the workloads above).
        if time.time() - T_START < float(os.environ.get('GEM_BENCH_RMAT_DEADLINE_S', '900')):
            # the same family at scale 20 (1M nodes / 15.4M edges): the largest power-law graph the sequential oracle has been run on (tests/golden/
            # n2v_ref_oracle_rmat20*_e16k.json) -- the parity point of this family; scale 22 below is its timing point
            a8 = copy.copy(args); a8.graph, a8.nodes, a8.edges = 'rmat', 1 << 20, 16000000
            extra['node2vec_rmat20'], w8 = time_workload('node2vec', a8, rank, world, comm, 1, 0, with_cpu=False)
            del w8
            torch.cuda.empty_cache()
            a6 = copy.copy(args); a6.graph, a6.nodes, a6.edges = 'rmat', 1 << 22, 64000000
            extra['node2vec_rmat22'], w6 = time_workload('node2vec', a6, rank, world, comm, 1, 0, with_cpu=False)
            del w6
            torch.cuda.empty_cache()
            a7 = copy.copy(a6)
            extra['gf_rmat22'], w7 = time_workload('gf', a7, rank, world, comm, 20, 2, with_cpu=False)
            del w7
            torch.cuda.empty_cache()
            if time.time() - T_START < float(os.environ.get('GEM_BENCH_RMAT_PQ_DEADLINE_S', '500')):
                # general (p, q) second-order walks at R-MAT scale (rejection sampling against the hubs' rows), (p, q) = (0.25, 4)
                a9 = copy.copy(a6); a9.ret_p, a9.inout_q = 0.25, 4.0
                extra['node2vec_rmat22_p0.25_q4'], w9 = time_workload('node2vec', a9, rank, world, comm, 1, 0, with_cpu=False)
                del w9
            else:
                extra['node2vec_rmat22_p0.25_q4'] = 'skipped: bench already ran %.0f s (GEM_BENCH_RMAT_PQ_DEADLINE_S); python bench.py --workload node2vec --graph rmat --nodes 4194304 --edges 64000000 --ret-p 0.25 --inout-q 4 --steps 1 --warmup 0' % (time.time() - T_START)
            for k in ('node2vec_rmat20', 'node2vec_rmat22', 'gf_rmat22'):
                extra[k]['cpu_baseline'] = {'value': None, 'kind': 'reference', 'note': 'not run: gem/c_exe/node2vec builds Sigma deg^2 second-order alias '
                                            'tables (max degree ~94k here) and exhausts host memory; gf.cpp per-edge rate: see gf_sbm1m_10m.cpu_baseline'}
        else:
            extra['rmat22_skipped'] = 'bench already ran %.0f s (GEM_BENCH_RMAT_DEADLINE_S)' % (time.time() - T_START)
        out['workloads'] = extra

    if rank == 0:
        emit(out)
    if world > 1:
        dist.barrier()                       # rank 0 scores the embedding after the timed region: nobody tears the group down under it
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
