#!/bin/bash
# GPU visit D of round 3: optimised front end + in-process multi-rank solve: parity suite, traces, bench.
set -u
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r03d
mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.build(); print('build ok')" > $OUT/build.log 2>&1; tail -2 $OUT/build.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -40 $OUT/pytest_gpu.log
MAVBA_FRONT_TRACE=$OUT/front_trace.txt timeout 300 python bench.py --steps 30 --warmup 6 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/bench_trace.log; echo "trace bench exit $?"
python scripts/_dbg/front_trace.py $OUT/front_trace.txt | tee $OUT/front_trace_summary.txt
rm -f $OUT/front_trace.txt
MAVBA_CHOL_TRACE=$OUT/chol_trace.txt timeout 300 python scripts/chol_trace.py C3 > $OUT/chol_trace_summary.txt 2>&1; head -90 $OUT/chol_trace_summary.txt
rm -f $OUT/chol_trace.txt
for C in C3 C2 C5; do
timeout 300 python bench.py --config $C --steps 60 --warmup 6 --no-cpu-baseline > $OUT/bench_$C.json 2> $OUT/bench_$C.log; grep "avg=" $OUT/bench_$C.log | head -6; python -c "import json;d=json.load(open('$OUT/bench_$C.json'));print(d['value'],d['ms_per_step'],d['reduced_system'])"
done
