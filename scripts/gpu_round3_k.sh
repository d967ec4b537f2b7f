#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r03k
mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.build(); print('build ok')" > $OUT/build.log 2>&1; tail -2 $OUT/build.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 -x > $OUT/pytest_quick.log 2>&1
echo "quick pytest exit $?"; tail -5 $OUT/pytest_quick.log
for C in C3 C2; do
timeout 300 python bench.py --config $C --steps 60 --warmup 6 --no-cpu-baseline > $OUT/bench_$C.json 2> $OUT/bench_$C.log; echo "== $C fused"; grep "avg=" $OUT/bench_$C.log | head -4; python -c "import json;d=json.load(open('$OUT/bench_$C.json'));print(d['value'],d['ms_per_step'],d['roofline']['kernel'],d['roofline']['frac'],d['jacobian_sweep']['frac'],d['jacobian_sweep']['avg_ms'])"
done
