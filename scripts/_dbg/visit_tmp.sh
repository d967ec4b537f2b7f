#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD; OUT=$PWD/gpurun_out/${1:-r04z}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q --tb=short -p no:cacheprovider --timeout 300 -k "dense_spd or small_system or dissection or envelope or full_solve or local_ba_windows or merged or termination" > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest.log
timeout 200 python scripts/_dbg/local_ba_latency.py 2>&1 | head -3
for c in C2 C3 C5; do
  timeout 400 python bench.py --config $c --steps 60 --warmup 6 --no-cpu-baseline > $OUT/bench_${c}.json 2> $OUT/bench_${c}.log
  python -c "
import json; d=json.loads(open('$OUT/bench_${c}.json').read().strip().splitlines()[-1]); k={r['kernel']:r['avg_ms'] for r in d['kernels']}; print('$c', d['value'], d['ms_per_step'], 'chol', k.get('chol_factor'), k.get('chol_backsolve'))"
done
