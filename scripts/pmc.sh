#!/bin/bash
# Collect PMC counters (own run, kernel-trace only) for: bash scripts/pmc.sh <tag> <python args...>
export TMPDIR=/tmp
R=$PWD; TAG=$1; shift
mkdir -p $R/gpurun_out/pmc
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_IFETCH SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "${PMC_EXTRA:-GRBM_GUI_ACTIVE}"; do
  i=$((i+1)); d=$R/gpurun_out/pmc/${TAG}_$i
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o pmc -- python "$@" > $d.log 2>&1); echo "pmc set $i rc $?"
done
find $R/gpurun_out/pmc -name "*kernel_trace.csv" -size +4M -delete
