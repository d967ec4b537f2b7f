#!/bin/bash
# GPU visit B of round 3: J-free front end - parity suite, A/B bench against the Jacobian-plane kernels, C2 / C5, kernel stats.
set -u
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r03b
mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.build(); print('build ok')" > $OUT/build.log 2>&1; tail -2 $OUT/build.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 900 -x > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -40 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $OUT/smoke.log
timeout 600 python bench.py --steps 60 --warmup 6 --no-cpu-baseline > $OUT/bench_C3.json 2> $OUT/bench_C3.log; echo "bench C3 exit $?"; grep "avg=" $OUT/bench_C3.log; python -c "import json;d=json.load(open('$OUT/bench_C3.json'));print(d['value'],d['ms_per_step'],d['solve'])"
MAVBA_FRONT_PLANES=1 timeout 600 python bench.py --steps 60 --warmup 6 --no-cpu-baseline > $OUT/bench_C3_planes.json 2> $OUT/bench_C3_planes.log; echo "bench C3 planes exit $?"; python -c "import json;d=json.load(open('$OUT/bench_C3_planes.json'));print(d['value'],d['ms_per_step'])"
for C in C2 C5; do
  timeout 600 python bench.py --config $C --steps 60 --warmup 6 --no-cpu-baseline > $OUT/bench_$C.json 2> $OUT/bench_$C.log; echo "bench $C exit $?"; grep "avg=" $OUT/bench_$C.log | head -8; python -c "import json;d=json.load(open('$OUT/bench_$C.json'));print(d['value'],d['ms_per_step'],d['solve'])"
done
rm -rf $OUT/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --steps 60 --warmup 6 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof_bench.log); echo "rocprof rc $?"
find $OUT/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/bench_C3_kernel_stats.csv \;
find $OUT/prof -name "*kernel_trace*.csv" -size +8M -delete
head -25 $OUT/bench_C3_kernel_stats.csv
