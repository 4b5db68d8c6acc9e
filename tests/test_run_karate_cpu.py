"""BASELINE configs[0] / north_star: "examples/run_karate.py and the evaluation suite run unchanged".  Checked HERE (the build container holds the
reference tree; the GPU box does not) with the reference's own script executed FROM WHERE IT LIES -- nothing of it is copied into this repository --
against the `gem` alias package: its imports by GEM's paths, its constructor kwargs, `learn_embedding(graph=, edge_f=, is_weighted=, no_python=)`,
`get_method_name()`, `evaluateStaticGraphReconstruction`, `viz.plot_embedding2D`, the SDNE stub's refusal.  This tier has no GPU, so the compute inside
`learn_embedding` is a stand-in built on the CPU oracle (tests may use it; the product never does), injected from OUTSIDE the script by a sitecustomize
module -- everything around it is the product's host code.  The same sequence on the HIP backend: tests/test_run_karate_gpu.py (our own driver
examples/run_karate_hip.py, same models and hyper-parameters)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

from conftest import golden_path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SCRIPT = '/root/reference/examples/run_karate.py'

STAND_IN = '''
import numpy as np
np.random.seed(7)
import oracle
from oracle import hope_oracle
from gem_amd.graph import edge_arrays
from gem_amd.embedding import gf, hope, lap, lle, node2vec as n2v


def _gf(self, graph=None, edge_f=None, is_weighted=False, no_python=True, **kw):
    if not graph:
        raise ValueError('graph needed')
    n, src, dst, w, _ = edge_arrays(graph)
    X0 = (0.01 * np.random.randn(n, self._d)).astype(np.float32)
    self._X = oracle.gf_train_f32(n, src, dst, w, self._d, self._eta, self._regu, min(int(self._max_iter), 2000), X0).astype(np.float64)
    return self._X


def _hope(self, graph=None, edge_f=None, is_weighted=False, no_python=False, **kw):
    if not graph:
        raise ValueError('graph needed')
    n, src, dst, w, order = edge_arrays(graph)
    self._X = np.asarray(hope_oracle.hope_dense(hope_oracle.adjacency(n, src, dst, w, order).toarray(), self._beta, self._d)[0], dtype=np.float64)
    return self._X


def _lap(self, graph=None, edge_f=None, is_weighted=False, no_python=False, **kw):
    n, src, dst, w, _ = edge_arrays(graph)
    self._X = np.asarray(hope_oracle.lap_eigmap_dense(n, src, dst, w, self._d)[0], dtype=np.float64)
    return self._X


def _lle(self, graph=None, edge_f=None, is_weighted=False, no_python=False, **kw):
    n, src, dst, w, _ = edge_arrays(graph)
    self._X = np.asarray(hope_oracle.lle_dense(n, src, dst, w, self._d)[0], dtype=np.float64)
    return self._X


def _n2v(self, graph=None, edge_f=None, is_weighted=False, no_python=False, **kw):
    if not graph:
        raise ValueError('graph needed')
    n, src, dst, w, _ = edge_arrays(graph)
    X, _ = oracle.n2v_train(n, src, dst, None, self._d, self._walk_len, self._num_walks, self._con_size, self._max_iter, float(self._ret_p), float(self._inout_p), 7, 11)
    self._X = X.astype(np.float64)
    return self._X


gf.GraphFactorization.learn_embedding = _gf
hope.HOPE.learn_embedding = _hope
lap.LaplacianEigenmaps.learn_embedding = _lap
lle.LocallyLinearEmbedding.learn_embedding = _lle
n2v.node2vec.learn_embedding = _n2v
'''


def test_reference_run_karate_runs_unchanged_against_the_alias_package(tmp_path):
    if not os.path.exists(REF_SCRIPT):
        pytest.skip('reference tree not present on this box')
    os.makedirs(tmp_path / 'data')
    shutil.copyfile(golden_path('karate.edgelist'), tmp_path / 'data' / 'karate.edgelist')
    (tmp_path / 'site').mkdir()
    (tmp_path / 'site' / 'sitecustomize.py').write_text(STAND_IN)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(tmp_path / 'site'), ROOT, os.environ.get('PYTHONPATH', '')]), MPLBACKEND='Agg')
    r = subprocess.run([sys.executable, REF_SCRIPT, '-node2vec', '1'], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    out = r.stdout
    blocks = re.findall(r'(\w+):\n\tTraining time: ([\d.]+)\n\tMAP: ([\d.eE+-]+) ', out)
    names = [b[0] for b in blocks]
    assert names == ['graph_factor_sgd', 'hope_gsvd', 'lap_eigmap_svd', 'lle_svd', 'node2vec_rw'], (names, out[-1500:], r.stderr[-2500:])
    assert out.count('Num nodes: 34, num edges: 77') == 6                     # the sixth model (SDNE) got as far as the header
    assert r.returncode != 0 and 'SDNE is out of scope' in r.stderr          # ... and refused to train (out of scope: SURVEY section 2)
    maps = {b[0]: float(b[2]) for b in blocks}
    assert all(0.0 < v < 1.0 for v in maps.values()), maps
