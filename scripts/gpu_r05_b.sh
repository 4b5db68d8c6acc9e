#!/bin/bash
# Round 5, GPU call B: Hogwild width / hot-threshold sweeps on R-MAT scale 17 (both layouts, 3 launches per setting, per-node APs saved for pairing with
# the sequential oracle on the build container) and the SGNS launch time against the wavefront cap on R-MAT scale 22.
mkdir -p gpurun_out/r05_rmat17 gpurun_out/r05_rmat22
( rocminfo | grep -i "compute unit\|marketing name\|max clock" | sort | uniq -c; rocm-smi --showcomputepartition --showmemorypartition 2>/dev/null | grep -v "^$" ) > gpurun_out/r05_box_b.txt 2>&1
python scripts/check_rmat17_launches.py --launches 0 --widths 0,1024,768,512,384,256,192,128 --width-layouts 27,11 --width-launches 3 --out gpurun_out/r05_rmat17 --tag _b --save-counts > gpurun_out/r05_rmat17_b.log 2>&1
python scripts/check_rmat17_launches.py --launches 0 --widths '' --hot-counts 4000,8000,16000,32000 --width-layouts 27 --width-launches 2 --out gpurun_out/r05_rmat17 --tag _bhot > gpurun_out/r05_rmat17_bhot.log 2>&1
python scripts/check_rmat17_launches.py --scale 22 --edges 64000000 --big 2048 --launches 0 --widths 0,1024,768,512,384,256 --width-layouts 27 --width-launches 1 --out gpurun_out/r05_rmat22 --tag _b --save-counts > gpurun_out/r05_rmat22_b.log 2>&1
tail -5 gpurun_out/r05_rmat17_b.log; tail -12 gpurun_out/r05_rmat22_b.log
