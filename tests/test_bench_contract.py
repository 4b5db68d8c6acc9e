"""CPU test: the bench lines committed under profiles/ carry every field of the bench.py contract."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('wl', ['node2vec', 'gf', 'hope'])
def test_committed_bench_line_has_the_contract_fields(wl):
    j = json.load(open(os.path.join(ROOT, 'profiles', 'r01_bench_%s.json' % wl)))
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in j, k
    assert j['vs_baseline'] is None and j['data'] == 'synthetic' and j['dtype'] == 'f32' and 'workload' in j['config']
    r = j['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['bound'] == 'hbm' and r['peak'] == 8000.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    c = j['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] in ('reference', 'port') and c['value'] > 0
    assert j['value'] > 20 * c['value']            # BASELINE target: >= 20x the CPU path
