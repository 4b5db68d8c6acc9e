#!/usr/bin/env python3
"""bench.py -- headline benchmark of the gem_amd hot path (see DESIGN.md "Measurement").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload gf|node2vec|hope]

One JSON line on stdout (rank 0).  A "step" is one pass of the hot path over the
synthetic graph that is already resident in HBM:
    gf        one SGD sweep over all edges of SBM(1M nodes, 10M edges), d=128
For N>1 launch with torch.distributed.run (one rank per GPU, RCCL): the path is sharded
by SOURCE NODE and the only collective is the all-gather of the owned row
blocks of the embedding table after each sweep.

`roofline.achieved` = algorithmic bytes per launch (3*4d+12 = 1548 B per
edge-update at d=128, times the updates one launch performs) / average launch duration,
measured here with HIP events on the launch stream.  `cpu_baseline` times the CPU oracle
port (gf.cpp:152-164 restated in C, single thread -- the reference loop is single threaded)
on a bounded number of sweeps of the same graph on this box's host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from gem_amd import _hip
from gem_amd.graph import sbm_graph, edge_arrays

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class GFWorkload(object):
    """BASELINE metric on the 1M-node SBM: edges/sec for Graph Factorization, d=128."""
    name = 'sbm1m_10m_gf_d128'
    metric = 'edges/sec'
    unit = 'edges/s'
    dtype = 'f32'
    kernel = 'gf_sweep_kernel'

    def __init__(self, args, rank, world):
        self.rank, self.world = rank, world
        self.n, self.m_target, self.blocks, self.d = args.nodes, args.edges, args.blocks, args.d
        self.eta, self.regu = 1e-2, 1e-2      # "trainable" setting; arithmetic per edge is identical
        t = time.time()
        g = sbm_graph(self.n, self.m_target, self.blocks, seed=20260923 + 4)
        self.n_edges = g.number_of_edges()
        n, src, dst, w, _ = edge_arrays(g)
        self.graph = (n, src, dst, w)
        # pad n so every rank owns an equal contiguous block of source rows
        self.n_pad = (n + world - 1) // world * world
        self.r0 = rank * (self.n_pad // world)
        self.r1 = min(self.r0 + self.n_pad // world, n)
        L = _hip.lib()
        self.plan = C.c_void_p()
        _hip.check(L.gemhip_gf_plan_create(n, len(src), _hip.ptr(src, C.c_int32), _hip.ptr(dst, C.c_int32), None, self.d,
                                           self.r0, max(self.r1, self.r0), C.byref(self.plan)))
        info = (C.c_int64 * 8)()
        _hip.check(L.gemhip_gf_plan_info(self.plan, info))
        self.updates, self.rows, self.levels = info[0], info[1], info[2]
        self.algo_bytes = info[5]
        dev = torch.device('cuda', torch.cuda.current_device())
        gen = torch.Generator(device=dev); gen.manual_seed(1234)
        self.Xa = (0.01 * torch.randn(self.n_pad, self.d, device=dev, generator=gen, dtype=torch.float32)).contiguous()
        self.Xb = self.Xa.clone()
        self.X = [self.Xa, self.Xb]
        _hip.check(L.gemhip_gf_plan_bind(self.plan, C.c_void_p(self.Xa.data_ptr()), C.c_void_p(self.Xb.data_ptr())))
        self.cur = 0
        self.L = L
        log('[rank %d] graph %d nodes %d edges, plan rows %d updates %d levels %d (setup %.1fs)' %
            (rank, n, self.n_edges, self.rows, self.updates, self.levels, time.time() - t))

    def step(self):
        s = torch.cuda.current_stream().cuda_stream
        _hip.check(self.L.gemhip_gf_plan_sweeps(self.plan, 1, self.eta, self.regu, C.c_void_p(s)))
        self.cur ^= 1
        if self.world > 1:
            new = self.X[self.cur]
            own = new[self.r0:self.r0 + self.n_pad // self.world].clone()
            dist.all_gather_into_tensor(new, own)

    def units_per_step(self):
        return self.n_edges            # graph.number_of_edges() per sweep

    def kernel_launches_per_step(self):
        return self.levels

    def cpu_baseline(self, budget_s=15.0):
        import oracle
        n, src, dst, w = self.graph
        X0 = (0.01 * np.random.RandomState(0).randn(n, self.d)).astype(np.float32)
        t = time.time(); oracle.gf_train_f32(n, src, dst, w, self.d, self.eta, self.regu, 1, X0); one = time.time() - t
        sweeps = max(1, min(20, int(budget_s / max(one, 1e-3))))
        t = time.time(); oracle.gf_train_f32(n, src, dst, w, self.d, self.eta, self.regu, sweeps, X0); el = time.time() - t
        return {'value': self.n_edges * sweeps / el, 'unit': self.unit, 'cores': 1, 'kind': 'port',
                'sample': '%d sweeps of the same %d-edge graph, oracle/gf_oracle.c (gf.cpp:152-164), 1 thread' % (sweeps, self.n_edges)}

    def check(self):
        x = self.X[self.cur]
        assert bool(torch.isfinite(x).all()), 'non-finite embedding'


WORKLOADS = {'gf': GFWorkload}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--workload', default=os.environ.get('GEM_BENCH_WORKLOAD', 'gf'), choices=sorted(WORKLOADS))
    ap.add_argument('--nodes', type=int, default=1000000)
    ap.add_argument('--edges', type=int, default=10000000)
    ap.add_argument('--blocks', type=int, default=100)
    ap.add_argument('--d', type=int, default=128)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)')
    torch.cuda.set_device(local)
    _hip.check(_hip.lib().gemhip_set_device(local))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
    if args.gpus != world and rank == 0:
        log('note: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE' % (args.gpus, world))

    wl = WORKLOADS[args.workload](args, rank, world)
    K = args.steps if args.steps is not None else 50
    W = args.warmup if args.warmup is not None else 5

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(W):
        wl.step()
    barrier()
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(K):
        wl.step()
    ev1.record()
    barrier()
    el = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    wl.check()
    t = torch.tensor([el], dtype=torch.float64, device='cuda')
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())

    if rank == 0:
        value = wl.units_per_step() * K / el
        launches = K * wl.kernel_launches_per_step()
        out = {
            'metric': wl.metric, 'value': value, 'unit': wl.unit, 'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': el * 1e3 / K, 'higher_is_better': True, 'scaling': 'strong' if world > 1 else 'weak',
            'vs_baseline': None, 'dtype': wl.dtype, 'data': 'synthetic',
            'config': {'workload': wl.name, 'nodes': args.nodes, 'directed_edges': wl.n_edges, 'd': args.d,
                       'sharding': 'source-node x%d' % world},
        }
        if world == 1:
            avg_launch_s = dev_ms * 1e-3 / launches
            achieved = wl.algo_bytes / wl.kernel_launches_per_step() / avg_launch_s / 1e9
            out['roofline'] = {'bound': 'hbm', 'kernel': wl.kernel, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                               'frac': achieved / HBM_PEAK_GBS, 'traffic': None,
                               'algorithmic_bytes_per_launch': wl.algo_bytes / wl.kernel_launches_per_step(),
                               'avg_launch_us': avg_launch_s * 1e6}
            if not args.no_cpu_baseline:
                out['cpu_baseline'] = wl.cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
