#!/bin/bash
# GPU visit F: device-side session set-up (radix sort, ordering on the device): tests, set-up timing host vs device.
set -u
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r03f
mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.build(); print('build ok')" > $OUT/build.log 2>&1; tail -2 $OUT/build.log
timeout 600 python -m pytest tests/test_gpu_setup.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 > $OUT/pytest_setup.log 2>&1
echo "setup tests exit $?"; tail -30 $OUT/pytest_setup.log
for M in host device; do
  MAVBA_SETUP=$M timeout 300 python scripts/_dbg/setup_timing.py C3 > $OUT/setup_$M.log 2>&1; echo "== $M"; grep -A40 "mavba_solve call 2" $OUT/setup_$M.log | head -60
done
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -15 $OUT/pytest_gpu.log
timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline > $OUT/bench_C3.json 2> $OUT/bench_C3.log; grep "avg=" $OUT/bench_C3.log | head -6; python -c "import json;d=json.load(open('$OUT/bench_C3.json'));print(d['value'],d['ms_per_step'],d['solve'])"
