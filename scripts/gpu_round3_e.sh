#!/bin/bash
# GPU visit E: front end with LDS-only barriers, persistent grid + next-tile prefetch.
set -u
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r03e
mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.build(); print('build ok')" > $OUT/build.log 2>&1; tail -2 $OUT/build.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_filter.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -x > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -5 $OUT/pytest_gpu.log
MAVBA_FRONT_TRACE=$OUT/front_trace.txt timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/bench_trace.log; echo "trace bench exit $?"
python scripts/_dbg/front_trace.py $OUT/front_trace.txt | tee $OUT/front_trace_summary.txt
rm -f $OUT/front_trace.txt
for G in 0 1024 2048 8192; do
  MAVBA_FRONT_GRID=$G timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline > $OUT/bench_grid$G.json 2> $OUT/bench_grid$G.log
  echo "grid $G: $(grep 'point_front ' $OUT/bench_grid$G.log | head -1) $(python -c "import json;print(json.load(open('$OUT/bench_grid$G.json'))['value'])")"
done
