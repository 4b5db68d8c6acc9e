#!/bin/bash
# Round 5, GPU call F: R-MAT scale 20 (the largest power-law graph the sequential oracle is being run on; 1M nodes / 15.4M edges, 432M tokens): Hogwild
# launches at the planner's width and across widths, both layouts, plus hot-row thresholds at fixed widths -- per-node APs saved for pairing with the
# oracle's run on the build container (scripts/pair_rmat_launches.py once tests/golden/n2v_ref_oracle_rmat20*_e16k.json exist).
mkdir -p gpurun_out/r05_rmat20
python scripts/check_rmat17_launches.py --scale 20 --edges 16000000 --launches 0 --widths 0,1536,1024,768,512,384,256 --width-layouts 27 --width-launches 2 --out gpurun_out/r05_rmat20 --tag _w27 --save-counts > gpurun_out/r05_rmat20_w27.log 2>&1
python scripts/check_rmat17_launches.py --scale 20 --edges 16000000 --launches 0 --widths 0,1536,768,384 --width-layouts 11 --width-launches 1 --out gpurun_out/r05_rmat20 --tag _w11 > gpurun_out/r05_rmat20_w11.log 2>&1
python scripts/check_rmat17_launches.py --scale 20 --edges 16000000 --launches 0 --widths '' --hot-counts 10000,20000 --hot-width 1024 --width-layouts 27 --width-launches 1 --out gpurun_out/r05_rmat20 --tag _h1024 > gpurun_out/r05_rmat20_h1024.log 2>&1
python scripts/check_rmat17_launches.py --scale 20 --edges 16000000 --launches 0 --widths '' --hot-counts 10000,20000 --hot-width 496 --width-layouts 27 --width-launches 1 --out gpurun_out/r05_rmat20 --tag _h496 > gpurun_out/r05_rmat20_h496.log 2>&1
grep -h '"mode"' gpurun_out/r05_rmat20_*.log | cut -c1-230
