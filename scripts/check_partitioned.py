"""Run under torch.distributed.run with N processes (on a 1-GPU box: GEM_BENCH_BACKEND=gloo, all ranks on cuda:0):
partitioned node2vec on the 16k-node SBM of tests/golden/n2v_ref_16k.json; rank 0 prints MAP and timing."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, torch.distributed as dist
from gem_amd import _hip, multi_gpu
from gem_amd.graph import sbm_graph, edge_arrays, to_csr
from gem_amd.embedding.node2vec import node2vec
from gem_amd.evaluation import reconstruction as gr
rank = int(os.environ.get('RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1))
torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)) % torch.cuda.device_count())
if world > 1:
    dist.init_process_group(os.environ.get('GEM_BENCH_BACKEND', 'nccl'), rank=rank, world_size=world)
ref = json.load(open(os.path.join(ROOT, 'tests/golden/n2v_ref_16k.json')))
p = ref['params']
g = sbm_graph(p['n'], p['edges'], p['blocks'], p['seed'])
n, src, dst, w, _ = edge_arrays(g)
row_ptr, col, ww = to_csr(n, src, dst, w)
b = multi_gpu.HipBackendN2V(n, row_ptr, col, ww, p['d'])
episodes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 11
job = multi_gpu.Node2VecPartitioned(b, multi_gpu.TorchComm(world), rank, world, n, p['num_walks'], p['walk_len'], p['window'], 1, seed=3,
                                    flags=flags, episodes=episodes)
t = time.time(); P = job.run(1.0, 1.0); torch.cuda.synchronize(); el = time.time() - t
if rank == 0:
    m = node2vec(d=p['d'], max_iter=1, walk_len=80, num_walks=10, con_size=10, ret_p=1, inout_p=1)
    MAP = gr.evaluateStaticGraphReconstruction(g, m, P.cpu().numpy().astype(np.float64), None)[0]
    print('world %d episodes %d flags %d: MAP %.4f (SNAP race-free %.4f, 8 threads %.4f)  %.2fs' %
          (world, episodes, flags, MAP, ref['snap']['t1']['MAP'], ref['snap']['t8']['MAP'], el), flush=True)
if world > 1:
    dist.destroy_process_group()
