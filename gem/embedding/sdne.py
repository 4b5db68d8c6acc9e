"""SDNE is OUT OF SCOPE of this backend (a Keras/TensorFlow auto-encoder, not on the hot path).
The class exists only so that GEM drivers which import it (examples/run_karate.py:18) still import; it constructs like
any StaticGraphEmbedding and refuses to train."""
from gem_amd.embedding.static_graph_embedding import StaticGraphEmbedding


class SDNE(StaticGraphEmbedding):
    hyper_params = {
        'method_name': 'sdne',
    }

    def __init__(self, *args, **kwargs):
        super(SDNE, self).__init__(*args, **kwargs)

    def learn_embedding(self, graph=None, edge_f=None, is_weighted=False, no_python=False, **_ignored):
        if not graph:
            raise ValueError('graph needed')
        raise NotImplementedError('SDNE is out of scope of the MI355X backend (HOPE, GraphFactorization, node2vec, '
                                  'LaplacianEigenmaps, LocallyLinearEmbedding are provided); use upstream GEM for SDNE')

    def get_edge_weight(self, i, j):
        raise NotImplementedError('SDNE is out of scope')
