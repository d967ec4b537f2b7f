#!/bin/bash
# GPU visit J: front end fused into the cluster kernel: parity suite, A/B bench.
set -u
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r03j
mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.build(); print('build ok')" > $OUT/build.log 2>&1; tail -2 $OUT/build.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_filter.py tests/test_gpu_setup.py tests/test_scene.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -x > $OUT/pytest_quick.log 2>&1
echo "quick pytest exit $?"; tail -15 $OUT/pytest_quick.log
for C in C3 C2; do
timeout 300 python bench.py --config $C --steps 60 --warmup 6 --no-cpu-baseline > $OUT/bench_$C.json 2> $OUT/bench_$C.log; echo "== $C fused"; grep "avg=" $OUT/bench_$C.log | head -7; python -c "import json;d=json.load(open('$OUT/bench_$C.json'));print(d['value'],d['ms_per_step'],d['solve']['iterations'],d['solve']['rmse_px'])"
MAVBA_NO_FUSE=1 timeout 300 python bench.py --config $C --steps 60 --warmup 6 --no-cpu-baseline > $OUT/bench_${C}_nofuse.json 2> $OUT/bench_${C}_nofuse.log; echo "== $C not fused"; python -c "import json;d=json.load(open('$OUT/bench_${C}_nofuse.json'));print(d['value'],d['ms_per_step'],d['solve']['iterations'],d['solve']['rmse_px'])"
done
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -12 $OUT/pytest_gpu.log
