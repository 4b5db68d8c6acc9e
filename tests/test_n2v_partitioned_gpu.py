"""GPU tests of the partitioned (multi-GPU episode) SGNS kernels and of the 16k-node parity point."""
import ctypes as C
import json

import numpy as np
import pytest
import torch

import oracle
from gem_amd import _hip, multi_gpu
from gem_amd.embedding.node2vec import node2vec
from gem_amd.evaluation import reconstruction as gr
from gem_amd.graph import edge_arrays, sbm_graph, to_csr
from conftest import golden_path

pytestmark = pytest.mark.gpu


def backend(G, d):
    n, src, dst, w, _ = edge_arrays(G)
    row_ptr, col, ww = to_csr(n, src, dst, w)
    return n, src, dst, multi_gpu.HipBackendN2V(n, row_ptr, col, ww, d)


def test_emit_pairs_is_the_trainmodel_pair_multiset(sbm1024):
    n, src, dst, b = backend(sbm1024, 16)
    b.walks(1.0, 1.0, 2, 80, 7, 11, 100, 1500)
    walks = np.empty((1400, 80), np.int32)
    _hip.check(b.L.gemhip_n2v_get_walks(b.h, _hip.ptr(walks, C.c_int32)))
    for lo, hi in ((0, 1400), (37, 411)):
        got = b.emit_pairs(10, 0, lo, hi, 7).cpu().numpy()
        want = oracle.sgns_pairs(walks[lo:hi], 10, 0, 100 + lo, 7)
        assert got.shape == want.shape
        key = lambda a: np.sort(a[:, 0].astype(np.int64) * n + a[:, 1])
        assert np.array_equal(key(got), key(want))
    # bucketed emission: same multiset, grouped by (context % parts, word % parts) with the reported bucket sizes
    for parts in (1, 3, 8):
        got, counts = b.emit_pairs_bucketed(10, 0, 0, 1400, 7, parts)
        got = got.cpu().numpy()
        want = oracle.sgns_pairs(walks, 10, 0, 100, 7)
        assert sum(counts) == len(want) == len(got)
        keyw = (want[:, 0] % parts) * parts + (want[:, 1] % parts)
        assert counts == np.bincount(keyw, minlength=parts * parts).tolist()
        keyg = np.repeat(np.arange(parts * parts), counts)                 # bucket of every output position
        glob = np.stack([got[:, 0] * parts + keyg // parts, got[:, 1] * parts + keyg % parts], axis=1)   # local rows -> global ids
        assert np.array_equal(key(glob), key(want))
    b.close()


@pytest.mark.parametrize('d,parts,flags', [(16, 4, 9), (128, 2, 11), (7, 3, 9)])
def test_train_pairs_deterministic_matches_oracle(sbm1024, d, parts, flags):
    n, src, dst, b = backend(sbm1024, d)
    b.walks(1.0, 1.0, 1, 40, 3, flags, 0, 256)
    b.vocab(); b.build_unigram_parts(parts)
    counts = b.counts.cpu().numpy()
    UT, KT, off = oracle.unigram_build_parts(counts, parts)
    UTd = np.empty(n, np.float32); KTd = np.empty(n, np.int32)
    _hip.check(b.L.gemhip_n2v_build_unigram_parts(b.h, parts, _hip.ptr(UTd, C.c_float), _hip.ptr(KTd, C.c_int32)))
    assert np.array_equal(UTd, UT) and np.array_equal(KTd, KT)
    pairs = b.emit_pairs(5, 0, 0, 256, 3)
    gi, gj = 1 % parts, 0                                     # bucket (context partition gi, word partition gj)
    sel = (pairs[:, 0] % parts == gi) & (pairs[:, 1] % parts == gj)
    bucket = (pairs[sel] // parts).contiguous()              # train_pairs takes local row indices
    assert bucket.shape[0] > 200
    P, N, _ = b.init_part_tables(3, gi, parts)
    Np = (0.05 * torch.randn(N.shape, generator=torch.Generator().manual_seed(1))).to(N.device)
    Po, No = P.cpu().numpy().copy(), Np.cpu().numpy().copy()
    b.train_pairs(bucket, gj, P, Np, 0.025, 0.01, 3, 77, flags | 4)
    torch.cuda.synchronize()
    oracle.sgns_train_pairs_local(bucket.cpu().numpy(), UT[off[gj]:off[gj + 1]], KT[off[gj]:off[gj + 1]], 0.025, 0.01, 3, 77, flags, Po, No)
    for got, want in ((P.cpu().numpy(), Po), (Np.cpu().numpy(), No)):
        assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max() + 1e-6
    b.close()


def test_partitioned_driver_world1_quality(sbm1024):
    """The episode schedule through the HIP backend (one rank = one partition): MAP equals the sequential algorithm's.
    On a 1024-node graph the pairs kernel runs 32 wavefronts wide (rows/32): Hogwild at that width costs 3-5 % of the MAP
    here (scripts/sweep_hogwild_waves.py; at 16k nodes the loss is 0.05 %, next test) -- hence the 8 % bar over 3 seeds."""
    n, src, dst, b = backend(sbm1024, 16)
    maps = []
    for seed in (1, 2, 3):
        job = multi_gpu.Node2VecPartitioned(b, multi_gpu.TorchComm(1), 0, 1, n, 10, 80, 10, 1, seed=seed, flags=9, episodes=16)
        P = job.run(1.0, 1.0).cpu().numpy().astype(np.float64)
        m = node2vec(d=16, max_iter=1, walk_len=80, num_walks=10, con_size=10, ret_p=1, inout_p=1)
        maps.append(gr.evaluateStaticGraphReconstruction(sbm1024, m, P, None)[0])
    Xs, _ = oracle.n2v_train(n, src, dst, None, 16, 80, 10, 10, 1, 1.0, 1.0, 1, 9)
    MAPs = gr.evaluateStaticGraphReconstruction(sbm1024, m, Xs.astype(np.float64), None)[0]
    assert abs(np.mean(maps) - MAPs) <= 0.08 * MAPs, (maps, MAPs)
    b.close()


def test_sbm16k_map_matches_race_free_snap():
    """SBM 16384 nodes / 164k edges, d=128: the real binary single-threaded reaches MAP 0.926 (8 threads: 0.289,
    tests/golden/n2v_ref_16k.json, scripts/make_golden_n2v_16k.py).  The HIP path (Hogwild, auto width) must match the
    race-free value within 1 %."""
    ref = json.load(open(golden_path('n2v_ref_16k.json')))
    p = ref['params']
    g = sbm_graph(p['n'], p['edges'], p['blocks'], p['seed'])
    m = node2vec(d=p['d'], max_iter=1, walk_len=p['walk_len'], num_walks=p['num_walks'], con_size=p['window'], ret_p=1, inout_p=1, seed=3)
    Y = m.learn_embedding(graph=g, is_weighted=True, no_python=True)
    MAP = gr.evaluateStaticGraphReconstruction(g, m, Y, None)[0]
    assert abs(MAP - ref['snap']['t1']['MAP']) <= 0.01 * ref['snap']['t1']['MAP'], (MAP, ref['snap'])
    assert MAP > ref['snap']['t8']['MAP']
