#!/bin/bash
# round 4, GPU call J: the slot table on a POWER-LAW graph (R-MAT scale 22, one node2vec pass): the library before the change against the shipped one, same box
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04j; mkdir -p $O
for lib in libgem_hip_r04base.so libgem_hip.so; do
  GEM_HIP_LIB=$PWD/gem_amd/$lib timeout 400 python bench.py --workload node2vec --graph rmat --nodes 4194304 --edges 64000000 --steps 1 --warmup 0 --no-cpu-baseline --no-api-wall 2> $O/bench_rmat22_$lib.log | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print(json.dumps({'lib': '$lib', 'workload': d['config']['workload'], 'ms_per_step': d['ms_per_step'], 'sgns_launch_us': d['roofline']['avg_launch_us'], 'pairs': d['roofline']['pairs_per_launch'], 'launch_plan': d['roofline']['launch_plan'], 'sampled_map': d['quality'].get('sampled_map')}))
" >> $O/ab_rmat22_slot_table.jsonl
done
cat $O/ab_rmat22_slot_table.jsonl
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
