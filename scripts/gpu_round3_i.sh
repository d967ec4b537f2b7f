#!/bin/bash
# GPU visit I: the round's evidence so far (bench lines, kernel stats, PMC passes) + host-thread experiment for the set-up.
set -u
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r03i
mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.build(); print('build ok')" > $OUT/build.log 2>&1; tail -2 $OUT/build.log
for T in 16 32 48; do
  MAVBA_HOST_THREADS=$T timeout 300 python scripts/_dbg/setup_timing.py C3 > $OUT/setup_threads$T.log 2>&1
  echo "threads $T: $(grep -A22 'mavba_solve call 2' $OUT/setup_threads$T.log | grep 'point clusters\|order blocks\|count terms\|flags\|finish_structure\|order on device\|session create' | tr -s ' ' | tr '\n' ';')"
done
bash scripts/measure_all.sh r03 > $OUT/measure_all.log 2>&1; tail -30 $OUT/measure_all.log
