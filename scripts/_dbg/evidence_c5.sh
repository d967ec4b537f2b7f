#!/bin/bash
# C5 evidence: chain trace of the persistent launch, HBM traffic and matrix-pipe counters of the C5 bench
set -u
export TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/${1:-r05}c5; mkdir -p $OUT
MAVBA_CHOL_TRACE=$OUT/raw.txt timeout 200 python scripts/chol_trace.py C5 > $OUT/chol_trace_C5.txt 2>&1; rm -f $OUT/raw.txt
grep -E "timing model|total forward|PRE_|TILE" $OUT/chol_trace_C5.txt
rm -rf $R/gpurun_out/pmc
bash scripts/pmc_traffic.sh --config C5 > $OUT/pmc_traffic.log 2>&1
python scripts/pmc_summary.py $R/gpurun_out/pmc $OUT/pmc_traffic_C5.json > $OUT/pmc_traffic_summary_C5.txt 2>&1
head -8 $OUT/pmc_traffic_summary_C5.txt
bash scripts/pmc_mfma.sh --config C5 > $OUT/pmc_mfma.log 2>&1
python scripts/pmc_mfma_summary.py $R/gpurun_out/pmc $OUT/pmc_mfma_C5.json > $OUT/pmc_mfma_summary_C5.txt 2>&1
cat $OUT/pmc_mfma_summary_C5.txt
rm -rf $R/gpurun_out/pmc
