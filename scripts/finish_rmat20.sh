#!/bin/bash
# After the two R-MAT scale 20 oracle runs (scripts/make_golden_n2v_scale.py --rmat-scale 20 --engine oracle --save-emb .refruns/oracle_rmat20_f{11,27}.npy)
# have finished: score them over the 16 384-node eligible sample (-> tests/golden/n2v_ref_oracle_rmat20*_e16k.json) and pair the per-node APs the GPU
# sweep saved (scripts/gpu_r05_f.sh) with them (-> profiles/r05_rmat20_width_sweep_paired.jsonl).
# NOTE: over 16 384 nodes the APs of this graph sum to 49 and one launch's paired gap has an s.e. of 1.4 %: the COMMITTED goldens are the 131 072-node ones
# (tests/golden/n2v_ref_oracle_rmat20*_e128k.json, scripts/score_oracle_ap.py --procs 4 on the same embeddings; on the common nodes both scorers agree
# exactly), the e16k files this script writes were an intermediate and are not kept.
set -e
cd "$(dirname "$0")/.."
for f in 11 27; do
  tag=oracle_rmat20_e16k; [ $f = 27 ] && tag=oracle_rmat20_vocab_order_e16k
  python scripts/make_golden_n2v_scale.py --nodes 1048576 --edges 16000000 --blocks 1 --seed 20260928 --rmat-scale 20 --engine oracle --flags $f \
      --eligible-sample 16384 --tag $tag --load-emb .refruns/oracle_rmat20_f$f.npy &
done
wait
rm -f tests/golden/n2v_ref_tmp_rmat20_f*.json
: > profiles/r05_rmat20_width_sweep_paired.jsonl
for t in _w27 _w11 _h1024 _h496; do python scripts/pair_rmat_launches.py gpurun_out/r05_rmat20 $t boxG 20 >> profiles/r05_rmat20_width_sweep_paired.jsonl; done
python - <<'PY'
import json, collections
agg = collections.OrderedDict()
for l in open('profiles/r05_rmat20_width_sweep_paired.jsonl'):
    r = json.loads(l)
    if 'gap_big_pct' not in r: continue
    k = (r['flags'], r.get('max_waves'), r.get('hot_count'))
    agg.setdefault(k, []).append((r['gap_big_pct'], r['gap_big_se_pct'], r['sgns_s']))
for k, v in agg.items():
    print('flags %s max_waves %s hot_count %s:' % k, ' '.join('%+.2f(%.2f)' % (a, b) for a, b, _ in v), ' %.1f s' % (sum(c for _, _, c in v) / len(v)))
PY
