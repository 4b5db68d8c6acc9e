"""`n_gpus` through the plugin API (backend knobs travel as constructor kwargs, gem/embedding/static_graph_embedding.py:14-19).

    GraphFactorization(d=128, ..., n_gpus=8)                 node2vec(d=128, ..., n_gpus=8, episodes=64)

`learn_embedding()` then shards the way north_star prescribes -- GF by source row with an all-gather of the owned row blocks per sweep, node2vec walks
by start node and SGNS over partitioned tables -- in one of two forms, chosen by how the process was started:

  * ONE process, n_gpus devices ("capi"): the library's own N-GPU entry points gemhip_gf_train_multi / gemhip_n2v_train_multi (include/gem_hip.h),
    RCCL loaded by the library.  `devices=[...]` names the ordinals (default 0 .. n_gpus-1); `virtual_ranks=True` runs the n_gpus ranks as virtual
    ranks on device 0 (the one-GPU test box: same sharding, schedule and kernels, collectives become copies).
  * one process PER GPU ("spmd"): torch.distributed is initialised with world_size == n_gpus (torch.distributed.run, backend "nccl" = RCCL).  Every
    rank calls learn_embedding() with the same arguments (collectively, like any torch.distributed program) and gets the full embedding back;
    gem_amd/multi_gpu.py's GFSharded / Node2VecPartitioned drive the same kernels through the same C ABI.  n_gpus='world' takes the world size.

n_gpus = 1 (the default) without `virtual_ranks` is the single-GPU path, untouched.  Errors: n_gpus that disagrees with an initialised process group,
more ranks than devices, or a workload that does not shard (HOPE: "replicas only") raise ValueError before anything is launched.
"""
import ctypes as C

import numpy as np

from gem_amd import _hip


def dist_world():
    """(torch.distributed module or None, rank, world) -- None unless a process group with more than one rank is initialised."""
    try:
        import torch.distributed as dist
    except ImportError:
        return None, 0, 1
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def resolve(model):
    """-> ('single' | 'capi' | 'spmd', n_gpus, devices list or None).  Pure host logic (tests/test_api_surface.py)."""
    n_gpus = getattr(model, '_n_gpus', 1)
    virt = bool(getattr(model, '_virtual_ranks', False))
    devices = getattr(model, '_devices', None)
    dist, _, world = dist_world()
    if n_gpus == 'world':
        n_gpus = world
    if n_gpus is None:
        n_gpus = 1
    if not isinstance(n_gpus, (int, np.integer)) or isinstance(n_gpus, bool) or n_gpus < 1:
        raise ValueError("n_gpus must be a positive integer or 'world', got %r" % (n_gpus,))
    n_gpus = int(n_gpus)
    if dist is not None and n_gpus > 1:
        if n_gpus != world:
            raise ValueError('n_gpus=%d but torch.distributed is initialised with world_size=%d (one process per GPU: pass n_gpus=%d or \'world\')'
                             % (n_gpus, world, world))
        if virt or devices is not None:
            raise ValueError('devices= / virtual_ranks= belong to the one-process form; under torch.distributed every rank uses its own current device')
        return 'spmd', n_gpus, None
    if devices is not None:
        devices = [int(v) for v in devices]
        if len(devices) != n_gpus:
            raise ValueError('devices names %d ordinals for n_gpus=%d' % (len(devices), n_gpus))
    if virt:
        if devices is not None and len(set(devices)) != 1:
            raise ValueError('virtual_ranks=True runs every rank on ONE device; devices=%r' % (devices,))
        devices = devices or [0] * n_gpus
    if n_gpus == 1 and not virt:
        return 'single', 1, devices
    return 'capi', n_gpus, devices


def device():
    """The torch device this rank computes on: its current GPU (torch.distributed.run sets one per process).  CPU only in the gloo tests, where the
    compute backend is a stand-in -- the HIP backends of gem_amd/multi_gpu.py refuse to run without a GPU (_hip.require_device)."""
    import torch
    return torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')


def broadcast_from_rank0(array):
    """numpy array -> rank 0's copy on every rank (seeds and numpy-drawn initial tables must agree across the processes)."""
    import torch
    dist, _, _ = dist_world()
    t = torch.from_numpy(np.ascontiguousarray(array)).to(device())
    dist.broadcast(t, 0)
    return t.cpu().numpy()


def _sync():
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def _devs(n_gpus, devices):
    return (C.c_int32 * n_gpus)(*devices) if devices is not None else None


def gf_capi(model, n, src, dst, w, X0, n_gpus, devices):
    """gemhip_gf_train_multi: X0 float32 (n, d) in, trained table out (in place).  Returns the stats dict."""
    st = (C.c_double * 8)()
    _hip.check(_hip.lib().gemhip_gf_train_multi(n, len(src), _hip.ptr(src, C.c_int32), _hip.ptr(dst, C.c_int32), _hip.ptr(_hip.as_f32(w), C.c_float),
                                                int(model._d), float(model._eta), float(model._regu), int(model._max_iter), n_gpus, _devs(n_gpus, devices),
                                                _hip.ptr(X0, C.c_float), st))
    return {'kernel_seconds': st[0], 'updates_per_sweep': st[1], 'rows_per_sweep': st[2], 'exchange_bytes_per_rank_per_sweep': st[3],
            'n_gpus': int(st[4]), 'virtual_ranks': bool(st[5]), 'driver': 'gemhip_gf_train_multi'}


def gf_spmd(model, n, src, dst, w, X0):
    """One process per GPU: this rank's source-row block through GFSharded (halo exchange or all-gather after every sweep -- bit-identical to one GPU),
    full table on every rank at the end.  X0 must be the SAME on every rank (seeded numpy draw)."""
    import torch
    from gem_amd import multi_gpu
    dist, rank, world = dist_world()
    dev = device()
    d = int(model._d)
    n_pad = (n + world - 1) // world * world
    Xa = torch.zeros(n_pad, d, device=dev, dtype=torch.float32)
    Xa[:n].copy_(torch.from_numpy(X0))
    Xb = Xa.clone()
    r0, r1 = rank * (n_pad // world), min((rank + 1) * (n_pad // world), n)
    b = multi_gpu.HipBackendGF(n, src, dst, w, d, r0, r1, Xa, Xb)
    try:
        job = multi_gpu.GFSharded(b, multi_gpu.TorchComm(world), rank, world, n, src, dst)
        table = Xa
        for _ in range(int(model._max_iter)):
            table = job.sweep(float(model._eta), float(model._regu))
        table = job.gather(table)
        _sync()
        X0[:] = table[:n].cpu().numpy()
    finally:
        b.close()
    return {'kernel_seconds': None, 'updates_per_sweep': b.updates, 'rows_per_sweep': b.rows, 'n_gpus': world, 'virtual_ranks': False,
            'driver': 'gem_amd.multi_gpu.GFSharded over torch.distributed (%s)' % dist.get_backend()}


def n2v_capi(model, n, row_ptr, col, ww, seed, flags, n_gpus, devices):
    X = np.empty((n, int(model._d)), dtype=np.float32)
    st = (C.c_double * 8)()
    _hip.check(_hip.lib().gemhip_n2v_train_multi(n, len(col), _hip.ptr(row_ptr, C.c_int64), _hip.ptr(col, C.c_int32), _hip.ptr(ww, C.c_float), int(model._d),
                                                 int(model._walk_len), int(model._num_walks), int(model._con_size), int(model._max_iter),
                                                 float(model._ret_p), float(model._inout_p), seed, flags, n_gpus, _devs(n_gpus, devices),
                                                 int(getattr(model, '_episodes', 64)), _hip.ptr(X, C.c_float), st))
    return X, {'walk_seconds': st[0], 'sgns_seconds': st[1], 'tokens': st[2], 'pairs': st[3], 'ring_bytes_per_rank_per_round': st[4], 'n_gpus': int(st[5]),
               'virtual_ranks': bool(st[6]), 'bucket_launches_per_rank': st[7], 'driver': 'gemhip_n2v_train_multi'}


def n2v_spmd(model, n, row_ptr, col, ww, seed, flags):
    import torch
    from gem_amd import multi_gpu
    dist, rank, world = dist_world()
    b = multi_gpu.HipBackendN2V(n, row_ptr, col, ww, int(model._d))
    b.vocab_order = bool(flags & _hip.N2V_VOCAB_ORDER)
    try:
        job = multi_gpu.Node2VecPartitioned(b, multi_gpu.TorchComm(world), rank, world, n, int(model._num_walks), int(model._walk_len), int(model._con_size),
                                            int(model._max_iter), seed=seed, flags=flags, episodes=int(getattr(model, '_episodes', 64)))
        P = job.run(float(model._ret_p), float(model._inout_p))
        _sync()
        X = P[:n].cpu().numpy().astype(np.float32)
        ph = (job.phase_seconds() if hasattr(b, 'event') else None) or {}
    finally:
        b.close()
    return X, {'pairs_by_this_rank': int(job.pairs_trained), 'n_gpus': world, 'virtual_ranks': False, 'phase_seconds': ph,
               'driver': 'gem_amd.multi_gpu.Node2VecPartitioned over torch.distributed (%s)' % dist.get_backend()}
