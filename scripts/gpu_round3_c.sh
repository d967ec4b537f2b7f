#!/bin/bash
# GPU visit C of round 3: where k_point_front and k_chol_persist spend their time (phase stamps), grid-cap experiment.
set -u
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r03c
mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.build(); print('build ok')" > $OUT/build.log 2>&1; tail -2 $OUT/build.log
MAVBA_FRONT_TRACE=$OUT/front_trace.txt timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/bench_trace.log; echo "trace bench exit $?"
python scripts/_dbg/front_trace.py $OUT/front_trace.txt | tee $OUT/front_trace_summary.txt
rm -f $OUT/front_trace.txt
for G in 512 1024 2048; do
  MAVBA_FRONT_GRID=$G timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline > $OUT/bench_grid$G.json 2> $OUT/bench_grid$G.log
  echo "grid $G: $(grep point_front $OUT/bench_grid$G.log | head -1)"
done
MAVBA_CHOL_TRACE=$OUT/chol_trace.bin timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > $OUT/bench_chol.json 2> $OUT/bench_chol.log; echo "chol trace exit $?"
python scripts/chol_trace.py $OUT/chol_trace.bin > $OUT/chol_trace_summary.txt 2>&1; head -80 $OUT/chol_trace_summary.txt
rm -f $OUT/chol_trace.bin
timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline > $OUT/bench_C3.json 2> $OUT/bench_C3.log; grep "avg=" $OUT/bench_C3.log | head -6; python -c "import json;d=json.load(open('$OUT/bench_C3.json'));print(d['value'],d['ms_per_step'],d['reduced_system'])"
