#!/bin/bash
# GPU visit N: full suite + the round's evidence set with the current head
set -u
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r03n
mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.build(); print('build ok')" > $OUT/build.log 2>&1; tail -1 $OUT/build.log
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $OUT/smoke.log
bash scripts/measure_all.sh r03 > $OUT/measure_all.log 2>&1; grep "rc" $OUT/measure_all.log
timeout 300 python scripts/_dbg/local_ba_latency.py > $OUT/small_call_latency.log 2>&1; sed -n 2,3p $OUT/small_call_latency.log
timeout 300 python scripts/_dbg/pose_latency.py >> $OUT/small_call_latency.log 2>&1
MAVBA_SETUP_TIMING=1 timeout 300 python scripts/_dbg/setup_timing.py C3 > $OUT/setup_C3.log 2>&1; grep -A32 "mavba_solve call 2" $OUT/setup_C3.log | grep "session create\|iterate\|end to end"
MAVBA_CHOL_TRACE=$OUT/chol_trace_C3.raw timeout 300 python scripts/chol_trace.py C3 > $OUT/chol_trace_C3.txt 2>&1; tail -4 $OUT/chol_trace_C3.txt
MAVBA_FUSED_TRACE=$OUT/fused_trace.raw timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2>&1; python scripts/_dbg/fused_trace.py $OUT/fused_trace.raw > $OUT/fused_trace_C3.txt; head -3 $OUT/fused_trace_C3.txt
rm -f $OUT/*.raw
