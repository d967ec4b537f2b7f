#!/bin/bash
# HBM traffic counters for bench.py kernels: separate --pmc passes (FETCH_SIZE costs 3 TCC slots,
# WRITE_SIZE 2), kernel-trace only. Usage: bash scripts/pmc_traffic.sh [bench args...]
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  d=$R/gpurun_out/pmc/traffic_$c
  rm -rf $d
  (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o pmc -- python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline "$@" > $d.log 2>&1); echo "pmc $c rc $?"
  find $d -name "*kernel_trace.csv" -size +4M -delete
done
