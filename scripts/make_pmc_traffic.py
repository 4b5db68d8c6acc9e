#!/usr/bin/env python3
"""profiles/<tag>_pmc_traffic.json from the PMC passes of scripts/profile_round4b.sh: FETCH_SIZE / WRITE_SIZE sums per kernel (scripts/pmc_summary.py
text, counters in KiB) x the calibration factors of profiles/r03_pmc_traffic.json (measured on launches of known byte counts in the kernels' own access
pattern; reproduced on two boxes) / the units of the run (pairs from the bench line printed under the same PMC pass, GF updates from its line).
    python scripts/make_pmc_traffic.py <dir with pmc_{n2v,gf}_{FETCH,WRITE}_SIZE.{txt,json}> <out.json>
A kernel whose passes are missing keeps the figures of the newest committed profile (so that bench.py, which replays the newest file, loses nothing)."""
import glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, out = sys.argv[1], sys.argv[2]
prev = {}
for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic.json'))):
    if os.path.abspath(f) != os.path.abspath(out):
        for k, v in json.load(open(f)).items():
            prev[k] = v
cal = prev['calibration']
ff, wf = cal['fetch_factor'], cal['write_factor']


def counter(path, kern):
    """(sum in KiB, dispatches) of the first line of a pmc_summary text that names `kern`"""
    if not os.path.exists(path):
        return None
    for line in open(path):
        if kern in line:
            m = re.search(r'dispatches=(\d+) \w+ = ([0-9.e+]+)', line)
            return float(m.group(2)), int(m.group(1))
    return None


def last_json(path):
    lines = [l for l in open(path).read().splitlines() if l.startswith('{')] if os.path.exists(path) else []
    return json.loads(lines[-1]) if lines else None


res = {'method': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (scripts/profile_round4b.sh); counters in KiB; the two '
                 'calibration factors of profiles/r03_pmc_traffic.json (launches of known byte counts in the same access pattern) applied to every kernel; '
                 'built by scripts/make_pmc_traffic.py', 'calibration': cal}
for k, v in prev.items():
    if k not in ('method', 'calibration'):
        res[k] = v
f, w, line = counter(os.path.join(src, 'pmc_n2v_FETCH_SIZE.txt'), 'sgns_win_kernel'), counter(os.path.join(src, 'pmc_n2v_WRITE_SIZE.txt'), 'sgns_win_kernel'), last_json(os.path.join(src, 'pmc_n2v_FETCH_SIZE.json'))
if f and w and line:
    pairs = line['roofline']['pairs_per_launch']
    fb, wb = f[0] / f[1] * 1024.0 * ff / pairs, w[0] / w[1] * 1024.0 * wf / pairs
    res['sgns_win_kernel'] = {'run': 'bench.py --workload node2vec --num-walks 2 --steps 1 (%d dispatches per pass: the timed launch and quality()\'s paired oracle launch)' % f[1],
                              'pairs': pairs, 'FETCH_SIZE_KB': f[0] / f[1], 'WRITE_SIZE_KB': w[0] / w[1], 'fetch_bytes_per_pair': fb, 'write_bytes_per_pair': wb,
                              'traffic_bytes_per_pair': fb + wb, 'algorithmic_bytes_per_pair': 7192,
                              'note': 'negative draws through the slot table (one 16-byte gather per draw, SgnsArgs::SK): round 3 / early round 4 measured 3 770 + 2 740 = 6 510 B per pair '
                                      'with two dependent gathers per draw'}
for kern in ('gf_sweep_rows_kernel',):
    f, w, line = counter(os.path.join(src, 'pmc_gf_FETCH_SIZE.txt'), kern), counter(os.path.join(src, 'pmc_gf_WRITE_SIZE.txt'), kern), last_json(os.path.join(src, 'pmc_gf_FETCH_SIZE.json'))
    if f and w and line:
        upd = None
        for wl in [line] + list((line.get('workloads') or {}).values() if isinstance(line.get('workloads'), dict) else (line.get('workloads') or [])):
            if isinstance(wl, dict) and wl.get('config', {}).get('nodes') == 1000000 and 'updates_per_sweep' in json.dumps(wl):
                upd = wl['roofline'].get('updates_per_sweep') or upd
        upd = upd or prev.get(kern, {}).get('updates_per_launch')
        fb, wb = f[0] / f[1] * 1024.0 * ff, w[0] / w[1] * 1024.0 * wf
        res[kern] = {'run': 'bench.py --workload gf --steps 10 --warmup 2 (SBM 1M/10M, 8 rows per wavefront, own-row loads non-temporal)', 'launches': f[1],
                     'FETCH_SIZE_KB_total': f[0], 'WRITE_SIZE_KB_total': w[0], 'fetch_bytes_per_launch': fb, 'write_bytes_per_launch': wb,
                     'traffic_bytes_per_launch': fb + wb, 'traffic_bytes_per_update': (fb + wb) / upd, 'updates_per_launch': upd}
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if 'bytes_per' in kk}) for k, v in res.items() if k in ('sgns_win_kernel', 'gf_sweep_rows_kernel')}))
