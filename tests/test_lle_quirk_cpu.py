"""gem/embedding/lle.py:28-30 normalises the adjacency IN PLACE (`normalize(A, norm='l1', axis=1, copy=False)`) and ignores the return
value.  sklearn can only do that for floating-point matrices: for a graph whose weights are integers (or absent) it normalises a float COPY
and the reference goes on with the un-normalised `I - A` -- an accident of dtypes, not the algorithm (Roweis & Saul's LLE needs rows that
sum to one, and the reference's own karate golden, loaded with float weights, is the normalised solution).  This backend always
normalises (gemhip_lle, hope.hip: "sklearn normalize(norm='l1')") and deliberately does NOT mirror the integer-dtype accident; this test
pins down what exactly is not mirrored."""
import networkx as nx
import numpy as np
import pytest
import scipy.sparse as sp

sklearn_pre = pytest.importorskip('sklearn.preprocessing')


def _adj(G):
    return sp.csr_matrix(nx.to_scipy_sparse_array(G.to_undirected()))


def test_reference_in_place_normalisation_is_a_no_op_for_integer_weights():
    G = nx.path_graph(5)
    for i, j in G.edges():
        G[i][j]['weight'] = 2                       # integer weights (a graph without weights gives an integer matrix as well)
    A = _adj(G)
    assert A.dtype.kind == 'i'
    before = A.toarray().copy()
    out = sklearn_pre.normalize(A, norm='l1', axis=1, copy=False)        # what lle.py:29 calls, return value dropped there
    assert np.array_equal(A.toarray(), before)                           # the reference's A is still un-normalised ...
    assert np.allclose(out.toarray().sum(axis=1), 1.0)                   # ... only the discarded copy was normalised


def test_float_weights_are_normalised_in_place_which_is_what_this_backend_implements():
    G = nx.path_graph(5)
    for i, j in G.edges():
        G[i][j]['weight'] = 2.0
    A = _adj(G)
    assert A.dtype.kind == 'f'
    sklearn_pre.normalize(A, norm='l1', axis=1, copy=False)
    assert np.allclose(A.toarray().sum(axis=1), 1.0)                     # lle.py:30 then forms I - (row-stochastic A): gemhip_lle's operator
