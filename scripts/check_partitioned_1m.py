"""Partitioned (episode) schedule at SBM 1M/10M on ONE GPU: kernel seconds per (virtual) rank, pairs and sampled MAP for N = parts virtual ranks
(Node2VecPartitioned.run_virtual: the rounds of an episode run rank after rank; buckets of a round touch disjoint rows) -- the per-GPU cost model of
the N-GPU path next to the single-GPU pass of bench.py.

    python scripts/check_partitioned_1m.py [parts ...]      (default: 1 2 4 8)      one JSON line per N
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from gem_amd import multi_gpu
from gem_amd.graph import sbm_graph, edge_arrays, to_csr
from gem_amd.evaluation import reconstruction as gr

parts_list = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
eps = int(os.environ.get('EPISODES', '64'))
g = sbm_graph(1000000, 10000000, 100, seed=20260923 + 4)              # the benchmark graph (bench.py), so the committed goldens apply
n, src, dst, w, _ = edge_arrays(g); row_ptr, col, ww = to_csr(n, src, dst, w)
b = multi_gpu.HipBackendN2V(n, row_ptr, col, ww, 128)
ref = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'n2v_ref_oracle_1000k_s4096.json')))
refs = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'n2v_ref_snap_1000k.json')))
nodes = np.random.RandomState(0).choice(n, size=4096, replace=False)
for parts in parts_list:
    job = multi_gpu.Node2VecPartitioned(b, multi_gpu.TorchComm(1), 0, 1, n, 10, 80, 10, 1, seed=20260923, flags=11, episodes=eps)
    torch.cuda.synchronize(); t = time.time()
    P = job.run_virtual(parts)
    torch.cuda.synchronize(); el = time.time() - t
    ap = gr.sampled_ap_gpu(g, None, P.cpu().numpy(), nodes)
    apo, aps = np.asarray(ref['ap']), np.asarray(refs['ap'])
    vs = job.virtual_rank_seconds
    do, ds = ap[:len(apo)] - apo, ap[:len(aps)] - aps
    print(json.dumps(dict(what='partitioned walk-ordered schedule (gemhip_sgns_train_part) on one GPU, SBM 1M/10M, d=128', virtual_ranks=parts, episodes=eps,
                          launches=eps * parts * parts, wall_seconds_all_ranks_serial=el, kernel_seconds_per_rank=vs, max_rank_seconds=max(vs),
                          pairs=int(job.pairs_trained), algorithmic_TBs_per_rank=job.pairs_trained / parts * 7192 / max(vs) / 1e12,
                          MAP=float(ap.mean()), nodes=len(ap),
                          vs_sequential_oracle_pct=float(100 * do.mean() / apo.mean()), vs_sequential_oracle_se_pct=float(100 * do.std(ddof=1) / np.sqrt(len(do)) / apo.mean()),
                          vs_snap_binary_pct=float(100 * ds.mean() / aps.mean()), vs_snap_binary_se_pct=float(100 * ds.std(ddof=1) / np.sqrt(len(ds)) / aps.mean()),
                          note='paired per node with the committed reference runs on the same graph and node sample; the draws of a bucket are those of the '
                               'single-GPU kernel per (walk, position), restricted to the partition tables')), flush=True)
    del P
