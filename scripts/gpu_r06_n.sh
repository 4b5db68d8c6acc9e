#!/bin/bash
# Round 6, GPU call N: where does a wavefront's time go on a power-law graph at the planner's width (one wavefront per CU: latency-bound)?  -DGEMHIP_SGNS_PROFILE
# phase cycles on R-MAT scale 20 at 256 and 768 wavefronts and on scale 22 at 548.
O=gpurun_out/r06n
mkdir -p $O
for cfg in "20 1:256" "20 1:768" "22 1:0"; do
  set -- $cfg
  GEM_HIP_LIB=$PWD/gem_amd/libgem_hip_prof.so timeout 900 python scripts/sweep_width_schedule.py --scale $1 --repeats 1 --out $O/prof.jsonl --schedules "$2" > $O/prof_$1_${2/:/_}.log 2>&1
  grep "sgns profile" $O/prof_$1_${2/:/_}.log
done
cat $O/prof.jsonl | cut -c1-300
