"""BASELINE configs[0] / north_star: "examples/run_karate.py and the evaluation suite run unchanged".  The reference's own script is executed unchanged,
from where it lies, in tests/test_run_karate_cpu.py (the build container holds the reference tree; this box does not, and nothing of the script is copied into
the repository).  HERE the same sequence -- the same five models with run_karate.py:47-53's hyper-parameters, imported by GEM's own paths through the `gem`
alias package, trained on the HIP backend, evaluated by evaluateStaticGraphReconstruction -- runs as this repository's own driver examples/run_karate_hip.py."""
import json
import os
import re
import subprocess
import sys

import pytest

from conftest import golden_path

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_run_karate_sequence_on_the_hip_backend(tmp_path):
    # the run is made repeatable from OUTSIDE the driver: a sitecustomize.py on PYTHONPATH seeds numpy's global RNG, which is where GraphFactorization's
    # 0.01*N(0,1) init (gf.py:92) and this backend's node2vec seed come from
    (tmp_path / 'site').mkdir()
    (tmp_path / 'site' / 'sitecustomize.py').write_text('import numpy as np\nnp.random.seed(7)\n')
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(tmp_path / 'site'), ROOT, os.environ.get('PYTHONPATH', '')]), MPLBACKEND='Agg')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'examples', 'run_karate_hip.py'), '-node2vec', '1', '-sdne', '1'], cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=600)
    out = r.stdout
    blocks = re.findall(r'(\w+):\n\tTraining time: ([\d.]+)\n\tMAP: ([\d.eE+-]+) ', out)
    names = [b[0] for b in blocks]
    assert names == ['graph_factor_sgd', 'hope_gsvd', 'lap_eigmap_svd', 'lle_svd', 'node2vec_rw'], (names, r.stderr[-2000:])
    assert out.count('Num nodes: 34, num edges: 77') == 6                     # the sixth model (SDNE) got as far as the header
    assert r.returncode != 0 and 'SDNE is out of scope' in r.stderr          # ... and refused to train
    maps = {b[0]: float(b[2]) for b in blocks}
    ref = json.load(open(golden_path('map_ref.json')))
    # HOPE: the embedding equals the reference's golden to 1e-7 (tests/test_hope_gpu.py), but its MAP on karate is numerically
    # fragile in the reference itself -- hope.py's rows follow insertion order while the evaluator indexes by node id, so the
    # ranking is decided by reconstructed weights of ~1e-7: the reference's golden embedding scores 0.103, a fresh reference run
    # 0.086 (tests/golden/map_ref.json), this backend 0.179 on an embedding that differs from the golden by 1e-7
    assert 0.5 * ref['karate_hope_fresh'] < maps['hope_gsvd'] < 0.3
    # GF and node2vec are randomly initialised / sampled.  Bands from the ORACLE's distribution under run_karate.py:47,53's
    # hyper-parameters over 200 numpy seeds (fp32 gf.cpp semantics / sequential TrainModel): GF MAP 0.484 +- 0.075 (min 0.272, max
    # 0.608; 7.5 % of seeds fall below 0.35, the band that made round 2's GPU tier red), node2vec 0.494 +- 0.038 (min 0.362, max 0.572).
    # GF band = mean -4.5 sd / +4 sd, node2vec = mean -5 sd / +4 sd: P(flake) < 1e-4 per assertion even UNSEEDED; with the seed above
    # both runs are deterministic (exact Gauss-Seidel sweeps; 34 nodes train on a single wavefront) -- oracle at seed 7: GF 0.492
    assert 0.15 < maps['graph_factor_sgd'] < 0.80
    assert 0.30 < maps['node2vec_rw'] < 0.65
    assert 0.0 < maps['lap_eigmap_svd'] < 1.0 and 0.0 < maps['lle_svd'] < 1.0
