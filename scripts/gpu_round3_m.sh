#!/bin/bash
# GPU visit M: phase stamps of the fused kernel
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03m; mkdir -p $OUT
MAVBA_FUSED_TRACE=$OUT/fused_trace.txt timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.log
python scripts/_dbg/fused_trace.py $OUT/fused_trace.txt | tee $OUT/fused_trace_summary.txt
