#!/bin/bash
# Round 5, GPU call H: SQ counters of the SGNS kernel on R-MAT scale 22 (768 wavefronts, hot rows) next to the SBM headline shape (one walk per node) --
# where the wave cycles go on the power-law graph.
PMC_GROUPS="1 2" bash scripts/pmc_passes.sh r05_sgns_rmat22 sgns_win -- python bench.py --workload node2vec --graph rmat --nodes 4194304 --edges 64000000 --num-walks 2 --steps 1 --warmup 0 --no-cpu-baseline --no-api-wall
PMC_GROUPS="1 2" bash scripts/pmc_passes.sh r05_sgns_sbm1m sgns_win -- python bench.py --workload node2vec --num-walks 2 --steps 1 --warmup 0 --no-cpu-baseline --no-api-wall
