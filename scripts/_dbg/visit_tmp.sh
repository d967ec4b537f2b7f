#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD; OUT=$PWD/gpurun_out/${1:-r04z}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -q --tb=short -p no:cacheprovider --timeout 300 -k "dense_spd or small_system or dissection or envelope or full_solve or random" > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest.log
for c in C2 C3 C5; do
  timeout 400 python bench.py --config $c --steps 60 --warmup 6 --no-cpu-baseline > $OUT/bench_${c}.json 2> $OUT/bench_${c}.log
  python -c "
import json; d=json.loads(open('$OUT/bench_${c}.json').read().strip().splitlines()[-1]); k={r['kernel']:r['avg_ms'] for r in d['kernels']}; print('$c', d['value'], d['ms_per_step'], 'chol', k.get('chol_factor'), k.get('chol_backsolve'))"
done
bash scripts/gpu_chol_trace.sh 2>&1 | grep -i "timing model"; grep -E "^ *(6|9|15|18|27|30|37|40|44) |total" gpurun_out/choltrace/summary_C3.txt; grep "timing model" gpurun_out/choltrace/summary_C*.txt
