#!/bin/bash
# GPU visit L: the round's evidence with the fused kernel (bench lines, kernel stats, PMC passes), small-call latency, full suite.
set -u
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r03l
mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.build(); print('build ok')" > $OUT/build.log 2>&1; tail -2 $OUT/build.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -8 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $OUT/smoke.log
bash scripts/measure_all.sh r03 > $OUT/measure_all.log 2>&1; tail -5 $OUT/measure_all.log
timeout 300 python scripts/_dbg/local_ba_latency.py > $OUT/small_call_latency.log 2>&1; head -12 $OUT/small_call_latency.log
timeout 300 python scripts/_dbg/pose_latency.py >> $OUT/small_call_latency.log 2>&1
MAVBA_SETUP_TIMING=1 timeout 300 python scripts/_dbg/setup_timing.py C3 > $OUT/setup_C3.log 2>&1; grep -A22 "mavba_solve call 2" $OUT/setup_C3.log | grep "session create\|iterate\|end to end\|order on device\|finish_structure"
