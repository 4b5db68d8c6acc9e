#!/bin/bash
# PMC passes (one counter group per run, --kernel-trace only, as MI355X_MICROARCH.md prescribes) over a command; per-kernel sums land in
# gpurun_out/pmc/<tag>_<group>.csv via scripts/pmc_summary.py.   usage: scripts/pmc_passes.sh <tag> <kernel-substring> -- <command...>
set -u
tag=$1; kern=$2; shift 3
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc
groups=(
 "SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU"
 "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_SMEM"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_HIT_sum TCC_MISS_sum"
)
i=0
for g in "${groups[@]}"; do
  if [ -n "${PMC_GROUPS:-}" ] && [[ " $PMC_GROUPS " != *" $i "* ]]; then i=$((i+1)); continue; fi
  d=gpurun_out/pmc/${tag}_g$i
  rm -rf "$d"
  timeout 600 rocprofv3 --kernel-trace --pmc $g --output-format csv -d "$d" -o run -- "$@" > "$d.log" 2>&1
  python scripts/pmc_summary.py "$d" "$kern" > gpurun_out/pmc/${tag}_g$i.txt 2>&1
  cat gpurun_out/pmc/${tag}_g$i.txt
  rm -rf "$d"
  i=$((i+1))
done
