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


def test_round3_default_line_is_one_run_with_every_baseline_config():
    """profiles/r03_bench_all.json = stdout of ONE un-profiled `python bench.py` on the GPU box (scripts/profile_round3.sh bench; nothing stitched):
    node2vec headline (BASELINE configs[3]) + GF configs[1] and 1M/10M + HOPE configs[2] and its directed (general Katz) variant + configs[4]
    (R-MAT scale 22: node2vec and GF).  Rooflines carry the calibrated real traffic next to the algorithmic rate; GF's cpu_baseline is the real
    gf.cpp binary (kind "reference") with the C port and GEM's Python loop beside it; node2vec's quotes the committed full-size reference runs."""
    j = json.load(open(os.path.join(ROOT, 'profiles', 'r03_bench_all.json')))
    _check_line(j)
    assert 'assembled_from' not in j
    assert j['config']['workload'].startswith('sbm1000k_10000k_node2vec')
    assert set(j['workloads']) == {'gf_sbm10k_100k_run_sbm_setting', 'gf_sbm1m_10m', 'hope_sbm100k_1m', 'hope_sbm100k_directed', 'node2vec_rmat22', 'gf_rmat22'}
    for name, w in j['workloads'].items():
        _check_line(w, with_cpu='rmat22' not in name)
        assert w['roofline']['achieved_traffic_GBs'] is None or 0 < w['roofline']['achieved_traffic_GBs'] < 8000.0
    assert j['workloads']['hope_sbm100k_directed']['roofline']['solver'] == 'block_krylov'
    assert j['workloads']['hope_sbm100k_1m']['roofline']['solver'] == 'symmetric_chebyshev_filter'
    g = j['workloads']['gf_sbm1m_10m']['cpu_baseline']
    assert g['kind'] == 'reference' and g['reference_binary']['loop_only_edges_per_s'] > 0 and g['python_loop']['edges_per_s'] > 0 and g['c_port']['edges_per_s'] > 0
    assert any(r['nodes'] == 1000000 for r in j['cpu_baseline']['committed_full_size_runs'])
    q = j['quality']
    assert q['oracle_map'] and abs(q['map_minus_oracle_map']) <= 0.01 * q['oracle_map']          # north_star: MAP within 1 % (paired with the sequential algorithm)
    assert 0.5 < j['roofline']['frac'] < 1.0 and j['roofline']['achieved_traffic_GBs'] < 8000.0
    # the launch the concurrency rule chose is part of the record: all seven wavefronts per CU on the headline graph, hot rows on the power-law one
    lp = j['roofline']['launch_plan']
    assert lp['concurrent_wavefronts'] == 1792 and lp['hot_rows'] == 0 and lp['rho'] < 0.015
    lr = j['workloads']['node2vec_rmat22']['roofline']['launch_plan']
    assert lr['hot_rows'] > 0 and lr['rho'] < 0.015 and lr['n_eff'] < lr['n_eff_cold']
    # HOPE: `value` is the device-resident solve; the numpy-out form GEM's API needs is reported beside it and is the same solve
    for name in ('hope_sbm100k_1m', 'hope_sbm100k_directed'):
        w = j['workloads'][name]
        pc = w['roofline']['pcie_inclusive']
        assert pc['outputs_identical_to_device_form'] and pc['seconds_per_step'] >= 1e-3 * w['ms_per_step'] * 0.98


def test_round4_default_line_carries_api_wall_and_honest_gf_fractions():
    """profiles/r04_bench_all.json = stdout of ONE un-profiled `python bench.py` of round 4: every roofline fraction is a fraction (GF's compulsory bytes,
    not the 1548-byte comparability figure that printed 1.6 in rounds 1-3), every BASELINE workload carries the `learn_embedding` API wall
    with its breakdown, the SNAP leg its own (text dump + binary + load), and the headline reports both parity legs."""
    j = json.loads(open(os.path.join(ROOT, 'profiles', 'r04_bench_all.json')).read().strip().splitlines()[-1])
    _check_line(j)
    assert j['config']['workload'].startswith('sbm1000k_10000k_node2vec')
    for name, w in [('headline', j)] + list(j['workloads'].items()):
        if not isinstance(w, dict) or 'roofline' not in w:
            continue
        assert 0.0 < w['roofline']['frac'] < 1.0, (name, w['roofline']['frac'])
        a = w['api_wall']
        for k in ('seconds', 'ingest_s', 'h2d_s', 'kernels_s', 'd2h_float64_s'):
            assert k in a and a[k] >= 0.0, (name, k)
        assert a['seconds'] >= a['kernels_s'] > 0.0
    g = j['workloads']['gf_sbm1m_10m']['roofline']
    assert g['kernel'] == 'gf_sweep_rows_kernel' and g['comparability_GBs'] > g['achieved'] and 0.7 < g['frac'] < 0.9
    assert 'regime' in j['workloads']['gf_sbm10k_100k_run_sbm_setting']['roofline']
    for leg in ('all_cores', 'single_thread_race_free'):
        assert j['cpu_baseline'][leg]['api_wall']['seconds'] >= j['cpu_baseline'][leg]['api_wall']['binary_s']
    q = j['quality']
    assert q['unigram_layout'].startswith('vocabulary order') and abs(q['map_minus_oracle_map']) <= 0.01 * q['oracle_map']
    assert abs(q['map_minus_reference_map']) <= 0.02 * q['reference_map']


def test_round4_closing_line():
    """profiles/r04b_bench_all.json = stdout of ONE un-profiled `python bench.py` at the end of round 4 (scripts/profile_round4b.sh bench): the contract
    fields, fractions that are fractions, the API wall in every BASELINE workload -- and the round's kernel work visible in the line itself: the SGNS
    traffic replayed from the slot-table PMC passes (6 054 B per pair), a headline no slower than the first session's, GF's sweep above 0.82 of the peak."""
    j = json.loads(open(os.path.join(ROOT, 'profiles', 'r04b_bench_all.json')).read().strip().splitlines()[-1])
    before = json.loads(open(os.path.join(ROOT, 'profiles', 'r04_bench_all.json')).read().strip().splitlines()[-1])
    _check_line(j)
    assert j['config']['workload'].startswith('sbm1000k_10000k_node2vec') and j['n_gpus'] == 1 and j['dtype'] == 'f32'
    r = j['roofline']
    assert r['kernel'] == 'sgns_win_kernel' and 0.75 < r['frac'] < 1.0 and r['frac'] > before['roofline']['frac']
    assert 'r04b_pmc_traffic.json' in r['traffic_source'] and abs(r['traffic'] / r['pairs_per_launch'] - 6054.3) < 1.0
    assert r['traffic'] < r['algorithmic_bytes_per_launch'] and j['value'] >= before['value']
    for name, w in [('headline', j)] + list(j['workloads'].items()):
        if not isinstance(w, dict) or 'roofline' not in w:
            continue
        assert 0.0 < w['roofline']['frac'] < 1.0, (name, w['roofline']['frac'])
        assert w['api_wall']['seconds'] >= w['api_wall']['kernels_s'] > 0.0, name
    g = j['workloads']['gf_sbm1m_10m']['roofline']
    assert g['kernel'] == 'gf_sweep_rows_kernel' and 0.82 < g['frac'] < 0.9
    q = j['quality']
    assert abs(q['map_minus_oracle_map']) <= 0.01 * q['oracle_map'] and abs(q['map_minus_reference_map']) <= 0.02 * q['reference_map']
    assert j['cpu_baseline']['kind'] == 'reference' and j['cpu_baseline']['value'] > 0


def test_round4_pmc_traffic_follows_from_the_committed_raw_counters():
    """profiles/r04b_pmc_traffic.json (what bench.py replays into roofline.traffic) is arithmetic on the committed counter sums: FETCH_SIZE / WRITE_SIZE in
    KiB per dispatch x 1024 x the calibration factor of the access pattern / the units of the launch (scripts/make_pmc_traffic.py)."""
    import re
    t = json.load(open(os.path.join(ROOT, 'profiles', 'r04b_pmc_traffic.json')))
    ff, wf = t['calibration']['fetch_factor'], t['calibration']['write_factor']

    def raw(path, kern, counter):
        for line in open(os.path.join(ROOT, 'profiles', path)):
            m = re.search(r'dispatches=(\d+) %s = ([0-9.e+]+)' % counter, line)
            if m and kern in line:
                return float(m.group(2)) / int(m.group(1))
        raise AssertionError((path, kern, counter))
    s = t['sgns_win_kernel']
    assert abs(raw('r04b_pmc_raw_counters_sgns.txt', 'sgns_win_kernel', 'FETCH_SIZE') * 1024 * ff / s['pairs'] - s['fetch_bytes_per_pair']) < 0.5
    assert abs(raw('r04b_pmc_raw_counters_sgns.txt', 'sgns_win_kernel', 'WRITE_SIZE') * 1024 * wf / s['pairs'] - s['write_bytes_per_pair']) < 0.5
    assert abs(s['fetch_bytes_per_pair'] + s['write_bytes_per_pair'] - s['traffic_bytes_per_pair']) < 1e-6 and s['traffic_bytes_per_pair'] < s['algorithmic_bytes_per_pair']
    g = t['gf_sweep_rows_kernel']
    assert abs(raw('r04b_pmc_raw_counters_gf.txt', 'gf_sweep_rows_kernel', 'FETCH_SIZE') * 1024 * ff / g['fetch_bytes_per_launch'] - 1.0) < 1e-3
    assert abs(raw('r04b_pmc_raw_counters_gf.txt', 'gf_sweep_rows_kernel', 'WRITE_SIZE') * 1024 * wf / g['write_bytes_per_launch'] - 1.0) < 1e-3
    # the GF sweep writes every row once: rows x 512 B at d = 128 (WRITE_SIZE is exact for this access pattern)
    assert abs(g['write_bytes_per_launch'] / (946188 * 512.0) - 1.0) < 1e-3



@pytest.mark.parametrize('record', ['r03_bench_all.json', 'r04_bench_all.json', 'r04b_bench_all.json'])
def test_compact_stdout_line_of_every_committed_full_record_fits_the_drivers_tail(record):
    """Round 4's default line was 23 KB and came back from the driver as `parsed: null`.  bench.py now prints the full record to stderr / a file and ONE
    compact line on stdout (bench.compact_line): for every full record committed so far it stays under 4 KB and keeps the contract fields, `roofline`
    and `cpu_baseline` with theirs, the parity gaps, and a summary of every other workload.  (tests/test_bench_gpu.py checks the same against real stdout.)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_for_test', os.path.join(ROOT, 'bench.py'))
    try:
        bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    except ImportError as e:                                   # bench.py imports torch at module level
        pytest.skip(str(e))
    full = json.loads(open(os.path.join(ROOT, 'profiles', record)).read().strip().splitlines()[-1])
    line = bench.compact_line(full, 'gpurun_out/bench_detail_latest.json')
    assert len(line) < 4096 and '\n' not in line
    j = json.loads(line)
    _check_line(j)
    assert j['value'] == pytest.approx(full['value'], rel=1e-6) and j['config']['workload'] == full['config']['workload']
    assert set(j['workloads']) == set(full['workloads'])
    for name, w in j['workloads'].items():
        if isinstance(w, dict) and 'frac' in w:
            assert w['frac'] == pytest.approx(full['workloads'][name]['roofline']['frac'], rel=1e-5)
    assert j['quality']['oracle_map'] == pytest.approx(full['quality']['oracle_map'], rel=1e-6)
    assert j['cpu_baseline']['kind'] == full['cpu_baseline']['kind'] and len(j['cpu_baseline']['sample']) <= 200
