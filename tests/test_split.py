"""CPU test: link-prediction split equals the reference's split_di_graph_to_train_test under the same numpy seed
(fixture tests/golden/split_ref.npz written by scripts/make_golden_split.py, which runs gem/utils/evaluation_util.py:39-53)."""
import numpy as np

from gem_amd.utils.evaluation_util import split_di_graph_to_train_test
from conftest import golden_path


def test_split_matches_reference(karate, sbm1024):
    ref = np.load(golden_path('split_ref.npz'))
    for name, G in (('karate', karate), ('sbm', sbm1024)):
        np.random.seed(17)
        tr, te = split_di_graph_to_train_test(G, 0.8, is_undirected=(name == 'sbm'))
        for tag, g in (('train', tr), ('test', te)):
            got = np.unique(g.src.astype(np.int64) * g.n + g.dst)
            assert np.array_equal(got, ref['%s_%s' % (name, tag)]), (name, tag)
