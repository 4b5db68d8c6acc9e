#!/usr/bin/env python3
"""Prints DESIGN.md section 5's round-3 table rows from profiles/r03_bench_all.json (so the document quotes the committed record, not memory)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
j = json.load(open(os.path.join(ROOT, 'profiles', sys.argv[1] if len(sys.argv) > 1 else 'r03_bench_all.json')))
w = j['workloads']
q = j['quality']
print('headline: %.3f M edges/s, %.2f s/pass, sgns %.2f s, algorithmic %.2f TB/s frac %.3f, real %.2f TB/s' % (j['value'] / 1e6, j['ms_per_step'] / 1e3, j['roofline']['avg_launch_us'] / 1e6,
      j['roofline']['achieved'] / 1e3, j['roofline']['frac'], j['roofline']['achieved_traffic_GBs'] / 1e3))
print('  quality: MAP %.4f (%d nodes); vs oracle %+.2f %% (se %.2f); vs reference %s' % (q['sampled_map'], q['nodes_sampled'], 100 * q['map_minus_oracle_map'] / q['oracle_map'],
      100 * q['map_minus_oracle_map_se'] / q['oracle_map'], ('%+.2f %% (se %.2f)' % (100 * q['map_minus_reference_map'] / q['reference_map'], 100 * q['map_minus_reference_map_se'] / q['reference_map'])) if q.get('reference_map') else None))
c = j['cpu_baseline']
print('  cpu: all cores %.0f edges/s MAP %.3f; 1 thread %.0f MAP %.3f; hip same sample %.3f' % (c['all_cores']['edges_per_s'], c['all_cores']['MAP'], c['single_thread_race_free']['edges_per_s'],
      c['single_thread_race_free']['MAP'], c['hip_map_same_sample']))
for k, x in w.items():
    if not isinstance(x, dict):
        print(k, x); continue
    r = x['roofline']; cb = x.get('cpu_baseline') or {}
    print('%s: value %.4g %s, %.4g ms/step, %s %.1f us, algorithmic %.2f TB/s frac %.3f, real %s TB/s' % (k, x['value'], x['unit'], x['ms_per_step'], r['kernel'], r['avg_launch_us'], r['achieved'] / 1e3, r['frac'],
          ('%.2f' % (r['achieved_traffic_GBs'] / 1e3)) if r.get('achieved_traffic_GBs') else None))
    if cb.get('reference_binary'):
        print('    gf.cpp loop %.1f M, e2e %.1f M; c port %.1f M; python loop %.2f M' % (cb['reference_binary']['loop_only_edges_per_s'] / 1e6, cb['reference_binary']['end_to_end_edges_per_s'] / 1e6,
              cb['c_port']['edges_per_s'] / 1e6, cb['python_loop']['edges_per_s'] / 1e6))
    elif cb.get('value'):
        print('    cpu %.4g %s' % (cb['value'], cb.get('kind')))
    if 'quality' in x:
        print('    quality', {kk: vv for kk, vv in x['quality'].items() if 'source' not in kk and 'note' not in kk})
