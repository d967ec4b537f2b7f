#!/bin/bash
# Quick GPU check while tuning: factorisation tests, C3 / C2 bench lines, local-window latency.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/quick; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "dense or chol or persistent or C2 or c2 or local or window" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
for c in C3 C2; do timeout 300 python bench.py --config $c --steps 60 --warmup 6 --no-cpu-baseline > $OUT/bench_$c.json 2> $OUT/bench_$c.log; grep -E "chol_factor|schur_fused|chol_backsolve" $OUT/bench_$c.log | head -3; python -c "
import json; d=json.loads(open('$OUT/bench_$c.json').read().strip().splitlines()[-1]); print('$c', d['value'], d['ms_per_step'])"; done
timeout 200 python scripts/_dbg/local_ba_latency.py 2>&1 | sed -n 2,6p
