#!/bin/bash
# round 4, GPU call F (second session): slot-table negatives (one gather per draw), GF non-temporal stores, HOPE fused Rayleigh-Ritz / Ritz rotation /
# two-pass column arg-max, hub-row Vose.  Parity tests of the touched paths first, then the A/Bs.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04f; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_n2v_gpu.py tests/test_hope_gpu.py tests/test_lap_gpu.py tests/test_gf_gpu.py tests/test_n2v_partitioned_gpu.py tests/test_multi_capi_gpu.py tests/test_rmat_gpu.py -q -m gpu -x 2>&1 | tail -40 ) > $O/pytest_touched.log 2>&1
tail -5 $O/pytest_touched.log
( GEM_AB_GOLDEN=n2v_ref_oracle_1000k_s4096.json GEM_HIP_LIB=$PWD/gem_amd/libgem_hip_r04base.so timeout 400 python scripts/ab_sgns_1m.py $O/ab_sgns_base.jsonl 1792:2:1 ) > $O/ab_sgns_base.out 2>&1
( GEM_AB_GOLDEN=n2v_ref_oracle_1000k_s4096.json timeout 500 python scripts/ab_sgns_1m.py $O/ab_sgns_slot_table.jsonl 1792:2:1 1792:1:1 1792:2:1 ) > $O/ab_sgns_slot_table.out 2>&1
cut -c1-400 $O/ab_sgns_base.jsonl $O/ab_sgns_slot_table.jsonl
( timeout 400 python scripts/ab_gf_rows.py 8/0 8/1 8/3 8/2 8/0 8/1 ) > $O/ab_gf_nt.jsonl 2> $O/ab_gf_nt.err
cat $O/ab_gf_nt.jsonl
( timeout 300 python scripts/ab_hope_sym.py two_pass:GEMHIP_HOPE_SYM_FUSED_RR=0 fused:GEMHIP_HOPE_SYM_FUSED_RR=1 two_pass_again:GEMHIP_HOPE_SYM_FUSED_RR=0 fused_again:GEMHIP_HOPE_SYM_FUSED_RR=1 ) > $O/ab_hope_fused.jsonl 2> $O/ab_hope_fused.err
cut -c1-420 $O/ab_hope_fused.jsonl
( GEMHIP_HOPE_COLMAX2=0 timeout 300 python scripts/ab_hope_sym.py one_block_per_column_argmax:GEMHIP_HOPE_SYM_FUSED_RR=1 ) > $O/ab_hope_colmax1.jsonl 2>> $O/ab_hope_fused.err
cut -c1-300 $O/ab_hope_colmax1.jsonl
( timeout 600 python scripts/time_alias_build.py 22 64000000 ) > $O/alias_build_rmat22.jsonl 2> $O/alias_build_rmat22.err
cat $O/alias_build_rmat22.jsonl; tail -3 $O/alias_build_rmat22.err
