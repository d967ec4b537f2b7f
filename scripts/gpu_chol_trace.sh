#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/choltrace; mkdir -p $OUT
for c in C3 C2; do MAVBA_CHOL_TRACE=$OUT/trace_$c.txt timeout 300 python scripts/chol_trace.py $c > $OUT/summary_$c.txt 2>&1; tail -45 $OUT/summary_$c.txt; done
