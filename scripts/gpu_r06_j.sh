#!/bin/bash
# Round 6, GPU call J: (1) why does RCCL's 1-rank init fail in a process that has not imported torch (calls H, I)?  (2) the north_star A/B microbenchmarks.
O=gpurun_out/r06j
mkdir -p $O
for mode in notorch torchfirst; do
  NCCL_DEBUG=INFO python - $mode > $O/rccl_$mode.log 2>&1 <<'PY'
import sys, ctypes as C
if sys.argv[1] == 'torchfirst':
    import torch
from gem_amd import _hip
L = _hip.lib()
n = C.c_int(); print('device_count rc', L.gemhip_device_count(C.byref(n)), n.value)
sec = C.c_double()
rc = L.gemhip_rccl_selftest(1, None, 1 << 20, C.byref(sec))
print('selftest rc', rc, L.gemhip_last_error().decode() if rc else 'ok', sec.value)
import subprocess, os
print(subprocess.run('grep -E "rccl|hsa|amdhip" /proc/%d/maps | awk "{print \$6}" | sort -u' % os.getpid(), shell=True, capture_output=True, text=True).stdout)
PY
  echo "== $mode"; grep -E "selftest rc|device_count|WARN|lib.*so" $O/rccl_$mode.log | head -20
done
scripts/microbench/northstar > $O/microbench_northstar.jsonl 2>&1; cat $O/microbench_northstar.jsonl
