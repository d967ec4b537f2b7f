#!/bin/bash
# One GPU-box visit: build check, parity tests, bench, rocprof kernel trace.
# Usage (from the repo root on the GPU box): bash scripts/gpu_check.sh [tests|bench|prof|all] ...
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
WHAT="${*:-all}"
has() { [[ " $WHAT " == *" $1 "* || " $WHAT " == *" all "* ]]; }

timeout 300 python -c "import __graft_entry__ as g; g.build(); print('build ok')" > $OUT/build.log 2>&1
tail -2 $OUT/build.log
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/device.log

if has tests; then
  timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 300 > $OUT/pytest_gpu.log 2>&1
  echo "pytest exit $?"; tail -40 $OUT/pytest_gpu.log
fi
if has smoke; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -5 $OUT/smoke.log
fi
if has bench2; then
  timeout 600 python bench.py --config C2 --steps 60 --warmup 6 > $OUT/bench_C2.json 2> $OUT/bench_C2.log
  echo "bench C2 exit $?"; tail -25 $OUT/bench_C2.log; cat $OUT/bench_C2.json
fi
if has bench; then
  timeout 900 python bench.py --steps 60 --warmup 6 > $OUT/bench_C3.json 2> $OUT/bench_C3.log
  echo "bench C3 exit $?"; tail -25 $OUT/bench_C3.log; cat $OUT/bench_C3.json
fi
if has prof; then
  rm -rf $OUT/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o bench -- python $OLDPWD/bench.py --steps 60 --warmup 6 --no-cpu-baseline > $OLDPWD/$OUT/prof_bench.json 2> $OLDPWD/$OUT/prof_bench.log)
  echo "rocprof exit $?"
  find $OUT/prof -name "*stats*" | head; find $OUT/prof -name "*kernel_stats*.csv" -exec head -30 {} \;
  # keep only the summaries (the raw trace can be large)
  find $OUT/prof -name "*kernel_trace*.csv" -size +8M -delete
fi
