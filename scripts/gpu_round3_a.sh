#!/bin/bash
# GPU visit A of round 3: the whole parity suite (incl. the new full-size C5 / sharded C3 tests), smoke, one bench line.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03a
mkdir -p $OUT
nproc > $OUT/nproc.txt; free -g > $OUT/mem.txt
timeout 300 python -c "import __graft_entry__ as g; g.build(); print('build ok')" > $OUT/build.log 2>&1; tail -2 $OUT/build.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 900 --durations=25 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -60 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $OUT/smoke.log
timeout 600 python bench.py --steps 60 --warmup 6 > $OUT/bench_C3.json 2> $OUT/bench_C3.log; echo "bench exit $?"; tail -30 $OUT/bench_C3.log; cat $OUT/bench_C3.json
