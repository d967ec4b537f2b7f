#!/bin/bash
# MFMA-utilisation counters for the reduced solve (k_chol_*) and the cluster kernel: their own rocprofv3 pass,
# kernel-trace only (never combined with --sys-trace / --stats). Usage: bash scripts/pmc_mfma.sh [bench args...]
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/pmc
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > $R/gpurun_out/pmc/mfma_counter_names.txt
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1)); d=$R/gpurun_out/pmc/mfma_$i
  rm -rf $d
  (cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o pmc -- python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline "$@" > $d.log 2>&1); echo "pmc mfma set $i rc $?"
  find $d -name "*kernel_trace.csv" -size +4M -delete
done
