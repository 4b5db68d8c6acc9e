"""Plugin base class of the HIP backend: same public surface as GEM's
StaticGraphEmbedding (gem/embedding/static_graph_embedding.py:5-83).

Kept behaviours:
  * ctor `Cls(*dicts, **kwargs)`: kwargs are merged into the CLASS-level
    `hyper_params` dict and every key becomes `self._<key>`; positional dicts are
    expanded the same way (static_graph_embedding.py:8-19).  The class-level
    mutation is deliberate -- the reference does it and its tests rely on
    `model.hyper_params['method_name'] == model.get_method_name()`.
    The knobs this backend ADDS (`backend_params`: seed, tolerances, ...) are not part
    of that contract and stay on the instance: a model built without them always gets
    the defaults, whatever an earlier model in the same process was given.
  * get_embedding() raises ValueError("Embedding not learned yet") before a fit (:38-46)
  * get_reconstructed_adj(X=None, node_l=None): stores X when given, zero
    diagonal (:48-65).  Here the n^2 Python loop is replaced by one matrix
    product supplied by the subclass (`_pair_matrix`); `get_edge_weight` stays
    the scalar definition and tests check both agree.
"""
import numpy as np


class StaticGraphEmbedding(object):
    hyper_params = {}
    backend_params = ('seed', 'device_init', 'flags', 'tol', 'oversample', 'krylov_steps', 'max_restarts', 'regroup_edges', 'verbose',
                      'n_gpus', 'devices', 'virtual_ranks', 'episodes')

    def __init__(self, *args, **kwargs):
        self._method_name = None
        self._d = None
        self._X = None
        self.hyper_params.update({name: value for name, value in kwargs.items() if name not in self.backend_params})
        for name, value in self.hyper_params.items():
            setattr(self, '_' + name, value)
        for name in self.backend_params:
            if name in kwargs:
                setattr(self, '_' + name, kwargs[name])
        for extra in args:
            for name in extra:
                setattr(self, '_' + name, extra[name])

    # ---- introspection -------------------------------------------------
    def get_method_name(self):
        return self._method_name

    def get_method_summary(self):
        return '%s_%d' % (self._method_name, self._d)

    def get_embedding(self):
        if self._X is None:
            raise ValueError("Embedding not learned yet")
        return self._X

    # ---- reconstruction ------------------------------------------------
    def _pair_matrix(self, X):
        """All-pairs get_edge_weight as one product; subclasses override when
        the pair score is not the plain inner product."""
        return X @ X.T

    def get_reconstructed_adj(self, X=None, node_l=None):
        if X is not None:
            self._X = X
        elif self._X is None:
            raise ValueError("Embedding not learned yet")
        # node_l is accepted and IGNORED, exactly like static_graph_embedding.py:48-65: the reference base class
        # reconstructs over all rows of the X it is given (evaluateStaticGraphReconstruction forwards node_l next to an
        # X the caller may already have sampled; only SDNE's override uses it)
        Xu = np.asarray(self._X, dtype=np.float64)
        adj = np.array(self._pair_matrix(Xu), dtype=np.float64)
        np.fill_diagonal(adj, 0.0)
        return adj

    # ---- to be provided by the method ------------------------------------
    def learn_embedding(self, graph=None, edge_f=None, is_weighted=False, no_python=False):
        raise NotImplementedError

    def get_edge_weight(self, i, j):
        raise NotImplementedError
