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


def _check_line(j, with_cpu=True):
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline'):
        assert k in j, k
    r = j['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    if with_cpu:
        c = j['cpu_baseline']
        assert c['kind'] in ('reference', 'port') and c['value'] > 0 and j['value'] > 20 * c['value']


def test_round2_default_line_carries_all_three_workloads():
    """The default `python bench.py` line of round 2 (profiles/r02_bench_all.json): node2vec headline + GF (run_sbm.py setting at
    10k/100k, and 1M/10M) + HOPE under "workloads", each with roofline and cpu_baseline; the node2vec baseline carries both SNAP runs
    (all cores = racy, one thread = race-free) with their MAPs, and the roofline says where its traffic figure comes from."""
    j = json.load(open(os.path.join(ROOT, 'profiles', 'r02_bench_all.json')))
    _check_line(j)
    assert j['config']['workload'].startswith('sbm1000k_10000k_node2vec')
    assert set(j['workloads']) == {'gf_sbm10k_100k_run_sbm_setting', 'gf_sbm1m_10m', 'hope_sbm100k_1m'}
    for w in j['workloads'].values():
        _check_line(w)
    c = j['cpu_baseline']
    assert c['all_cores']['MAP'] < c['single_thread_race_free']['MAP'] and abs(c['hip_map_same_sample'] - c['single_thread_race_free']['MAP']) < 0.02
    assert 'traffic_source' in j['roofline']
    assert j['quality']['nodes_sampled'] == 1024
