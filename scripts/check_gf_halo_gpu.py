"""GF row sharding with the halo exchange through the HIP backend == one GPU, bit for bit.
Launch:  GEM_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
             --master-port 29533 scripts/check_gf_halo_gpu.py      (ranks share cuda:0 on a 1-GPU box; "nccl" on a real node)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from gem_amd import _hip, multi_gpu
from gem_amd.graph import edge_arrays, sbm_graph

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
local = int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count()
torch.cuda.set_device(local); _hip.check(_hip.lib().gemhip_set_device(local))
dist.init_process_group(os.environ.get('GEM_BENCH_BACKEND', 'nccl'), rank=rank, world_size=world)
comm = multi_gpu.TorchComm(world)
g = sbm_graph(30001, 300000, 12, seed=5)
n, src, dst, w, _ = edge_arrays(g)
d, sweeps = 64, 6
dev = torch.device('cuda', local)
n_pad = (n + world - 1) // world * world
X0 = torch.zeros(n_pad, d, device=dev); X0[:n] = torch.from_numpy((0.05 * np.random.RandomState(1).randn(n, d)).astype(np.float32)).to(dev)


def run(r0, r1, job_world, job_rank, edges):
    Xa, Xb = X0.clone(), X0.clone()
    b = multi_gpu.HipBackendGF(n, src, dst, None, d, r0, r1, Xa, Xb)
    job = multi_gpu.GFSharded(b, comm if job_world > 1 else multi_gpu.TorchComm(1), job_rank, job_world, n, *(edges or ()))
    for _ in range(sweeps):
        last = job.sweep(0.02, 0.01)
    last = job.gather(last).clone()
    b.close()
    return last, job


blk = n_pad // world
sharded, job = run(rank * blk, min((rank + 1) * blk, n), world, rank, (src, dst))
single, _ = run(0, n, 1, 0, None)
torch.cuda.synchronize()
same = bool(torch.equal(sharded[:n], single[:n]))
print('[rank %d] halo plan %s rows per rank %s; sharded == single GPU: %s' % (rank, 'used' if job.halo else 'fallback', job.halo_rows, same), flush=True)
assert same and job.halo
dist.barrier()
dist.destroy_process_group()
