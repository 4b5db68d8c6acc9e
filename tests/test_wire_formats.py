"""Wire formats of GEM's native executables (gem/utils/graph_util.py:129-169) and their array fast paths:
the fast writers are byte-identical to the per-line ones, the fast readers return what the per-line ones return."""
import os

import numpy as np
import pytest

from gem_amd.graph import EdgeListGraph, sbm_graph
from gem_amd.utils import graph_util as gu


def test_edge_list_writers_are_byte_identical(tmp_path, karate):
    for header, slow in ((True, gu.saveGraphToEdgeListTxt), (False, gu.saveGraphToEdgeListTxtn2v)):
        a, b = str(tmp_path / 'a.txt'), str(tmp_path / 'b.txt')
        slow(karate, a)
        gu.saveEdgeListArrays(karate, b, header=header)
        assert open(a).read() == open(b).read()
    g = sbm_graph(500, 4000, 5, seed=1)
    g = EdgeListGraph(g.n, g.src, g.dst, np.random.RandomState(0).rand(len(g.src)).astype(np.float32) * 3)
    a, b = str(tmp_path / 'a.txt'), str(tmp_path / 'b.txt')
    gu.saveGraphToEdgeListTxt(g, a)
    gu.saveEdgeListArrays(g, b, header=True)
    assert open(a).read() == open(b).read()


def test_edge_list_reader_matches_reference_parser(tmp_path, karate):
    f = str(tmp_path / 'g.txt')
    gu.saveGraphToEdgeListTxtn2v(karate, f)
    G = gu.loadGraphFromEdgeListTxt(f)
    g = gu.loadEdgeListArrays(f)
    assert g.number_of_edges() == G.number_of_edges()
    assert sorted(zip(g.src.tolist(), g.dst.tolist())) == sorted(G.edges())
    assert np.all(g.w == 1.0)
    gu.saveGraphToEdgeListTxt(karate, f)                       # gf format: n, m header
    g2 = gu.loadEdgeListArrays(f, header=True)
    assert g2.n == len(karate.nodes) and np.array_equal(g2.src, g.src) and np.array_equal(g2.dst, g.dst)
    with open(f, 'a') as fh:
        fh.write('0 1 1.000000\n')
    with pytest.raises(ValueError):
        gu.loadEdgeListArrays(f, header=True)                  # header/edge count mismatch
    two = str(tmp_path / 'two.txt')
    open(two, 'w').write('0 3\n2 1\n')
    g3 = gu.loadEdgeListArrays(two)
    assert g3.n == 4 and g3.w is None and g3.src.tolist() == [0, 2]
    with pytest.raises(ValueError):
        gu.loadEdgeListArrays(two, n=3)
    empty = str(tmp_path / 'empty.txt')
    open(empty, 'w').write('')
    assert gu.loadEdgeListArrays(empty).number_of_edges() == 0


def test_embedding_text_round_trip_and_id_placement(tmp_path):
    X = np.random.RandomState(0).randn(50, 7)
    f = str(tmp_path / 'x.emb')
    gu.saveEmbedding(X, f)
    A, B = gu.loadEmbedding(f), gu.loadEmbeddingFast(f)
    assert np.array_equal(A, B) and np.allclose(A, X, rtol=1e-7)
    # the executables write rows in their own order and omit nodes they never saw: rows are placed by id
    open(f, 'w').write('4 2\n3 1.5 -2\n0 0.25 7\n')
    A, B = gu.loadEmbedding(f), gu.loadEmbeddingFast(f)
    assert np.array_equal(A, B) and A[3].tolist() == [1.5, -2.0] and A[0].tolist() == [0.25, 7.0] and not A[1:3].any()
    open(f, 'w').write('4 2\n')
    assert not gu.loadEmbeddingFast(f).any()
    open(f, 'w').write('4 3\n0 1 2\n')
    with pytest.raises(ValueError):
        gu.loadEmbeddingFast(f)


def test_embedding_binary_container(tmp_path):
    f = str(tmp_path / 'x.gemb')
    for dt in (np.float32, np.float64):
        X = np.random.RandomState(1).randn(33, 5).astype(dt)
        gu.saveEmbeddingBinary(X, f)
        assert os.path.getsize(f) == 32 + X.nbytes
        Y = gu.loadEmbeddingBinary(f)
        assert Y.dtype == dt and np.array_equal(X, Y)
        assert np.array_equal(np.asarray(gu.loadEmbeddingBinary(f, mmap=True)), X)
    with open(f, 'r+b') as fh:
        fh.truncate(40)
    with pytest.raises(ValueError):
        gu.loadEmbeddingBinary(f)
    open(f, 'wb').write(b'not an embedding at all, just thirty-two bytes..')
    with pytest.raises(ValueError):
        gu.loadEmbeddingBinary(f)
    with pytest.raises(ValueError):
        gu.saveEmbeddingBinary(np.zeros(3), f)
