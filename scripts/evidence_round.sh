#!/bin/bash
# One GPU-box visit for a round's committed evidence beyond scripts/measure_all.sh: latencies, set-up phases, phase traces.
#   bash scripts/evidence_round.sh r05      (results under gpurun_out/<tag>/, copied to profiles/<tag>_* by the caller)
set -u
export TMPDIR=/tmp
TAG=${1:-r04}
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
bash scripts/measure_all.sh $TAG > $OUT/measure_all.log 2>&1; tail -5 $OUT/measure_all.log
{ timeout 200 python scripts/_dbg/local_ba_latency.py; timeout 200 python scripts/_dbg/c1_latency.py; timeout 200 python scripts/_dbg/pose_latency.py; } > $OUT/small_call_latency.log 2>&1
timeout 200 python scripts/_dbg/setup_timing.py C3 > $OUT/setup_timing_C3.log 2>&1
timeout 300 python scripts/_dbg/shim_timing.py C3 real > $OUT/shim_dropin_C3.log 2>&1
timeout 300 python scripts/_dbg/scene_timing.py > $OUT/scene_timing_C3.log 2>&1
MAVBA_ROWS_TRACE=$OUT/rows_trace_raw.txt timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
{ python scripts/_dbg/rows_trace.py $OUT/rows_trace_raw.txt 1; python scripts/_dbg/rows_trace.py $OUT/rows_trace_raw.txt 0; python scripts/_dbg/rows_timeline.py $OUT/rows_trace_raw.txt; } > $OUT/rows_trace_C3.txt 2>&1; rm -f $OUT/rows_trace_raw.txt
MAVBA_CHOL_TRACE=$OUT/chol_trace_raw.txt timeout 300 python scripts/chol_trace.py C3 > $OUT/chol_trace_C3.txt 2>&1; rm -f $OUT/chol_trace_raw.txt
rm -rf $OUT/tl; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o w -- python $R/scripts/_dbg/window_timeline.py > /dev/null 2>&1)
python scripts/_dbg/iter_timeline.py $OUT/tl > $OUT/window_timeline.txt 2>&1; rm -rf $OUT/tl
rm -rf $OUT/tl3; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl3 -o b -- python $R/bench.py --steps 40 --warmup 6 --no-cpu-baseline > /dev/null 2>&1)
python scripts/_dbg/iter_timeline.py $OUT/tl3 first > $OUT/iteration_timeline_C3.txt 2>&1; rm -rf $OUT/tl3
timeout 60 scripts/_dbg/pipe_bench > $OUT/pipe_bench_fp64.txt 2>&1
timeout 60 scripts/_dbg/issue_bench > $OUT/issue_bench.txt 2>&1
MAVBA_CHOL_TRACE=$OUT/chol_trace_raw.txt timeout 300 python scripts/chol_trace.py C2 > $OUT/chol_trace_C2.txt 2>&1; rm -f $OUT/chol_trace_raw.txt
ls -la $OUT
timeout 200 python scripts/_dbg/window_setup.py > $OUT/window_setup.log 2>&1
