#!/bin/bash
# round 4, GPU call G: GF non-temporal hints (own-row load is the new default; + the CSR streams, + rows per wavefront re-tuned under it), HOPE SpMM with
# the oldest recurrence term read non-temporally, the de-flaked R-MAT-17 test, PMC traffic of the SGNS kernel with the slot table
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04g; mkdir -p $O
( timeout 400 python scripts/ab_gf_rows.py 8/0 8/2 8/6 8/4 4/2 16/2 8/2 8/6 ) > $O/ab_gf_nt2.jsonl 2> $O/ab_gf_nt2.err
cat $O/ab_gf_nt2.jsonl
( GEMHIP_HOPE_SPMM_NT=0 timeout 200 python scripts/ab_hope_sym.py w2_plain: ) > $O/ab_hope_spmm_nt.jsonl 2> $O/ab_hope_spmm_nt.err
( GEMHIP_HOPE_SPMM_NT=1 timeout 200 python scripts/ab_hope_sym.py w2_nontemporal: ) >> $O/ab_hope_spmm_nt.jsonl 2>> $O/ab_hope_spmm_nt.err
( GEMHIP_HOPE_SPMM_NT=0 timeout 200 python scripts/ab_hope_sym.py w2_plain_again: ) >> $O/ab_hope_spmm_nt.jsonl 2>> $O/ab_hope_spmm_nt.err
( GEMHIP_HOPE_SPMM_NT=1 timeout 200 python scripts/ab_hope_sym.py w2_nontemporal_again: ) >> $O/ab_hope_spmm_nt.jsonl 2>> $O/ab_hope_spmm_nt.err
grep -v download $O/ab_hope_spmm_nt.jsonl | cut -c1-330
( timeout 600 python -m pytest tests/test_rmat_gpu.py tests/test_gf_gpu.py -q -m gpu 2>&1 | tail -8 ) > $O/pytest_rmat_gf.log 2>&1
tail -4 $O/pytest_rmat_gf.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_n2v_$c -o run -- python bench.py --workload node2vec --num-walks 2 --steps 1 --warmup 0 --no-cpu-baseline --no-api-wall > $O/pmc_n2v_$c.json 2> $O/pmc_n2v_$c.log
  python scripts/pmc_summary.py $O/pmc_n2v_$c sgns > $O/pmc_n2v_$c.txt; cat $O/pmc_n2v_$c.txt; rm -rf $O/pmc_n2v_$c
done
