#!/bin/bash
# tuning visit: merged launches of the local window
set -u
export TMPDIR=/tmp
R=$PWD; OUT=$PWD/gpurun_out/${1:-r04w}; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q --tb=short -p no:cacheprovider --timeout 300 -k "merged or local_ba_windows or small_system or stepwise or termination_kinds or lm_decision" > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -15 $OUT/pytest.log
for m in 1 0; do
  echo "== MAVBA_MERGE=$m"
  MAVBA_MERGE=$m timeout 200 python scripts/_dbg/local_ba_latency.py 2>&1 | head -3
done
timeout 100 python scripts/_dbg/window_setup.py 2>&1 | tail -32 > $OUT/window_setup.txt
rm -rf $OUT/tl; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o w -- python $R/scripts/_dbg/window_timeline.py > /dev/null 2>&1)
python scripts/_dbg/iter_timeline.py $OUT/tl > $OUT/window_timeline.txt 2>&1; rm -rf $OUT/tl
cat $OUT/window_timeline.txt
for c in C2 C3; do
  timeout 400 python bench.py --config $c --steps 60 --warmup 6 --no-cpu-baseline > $OUT/bench_$c.json 2> $OUT/bench_$c.log
  python -c "
import json; d=json.loads(open('$OUT/bench_$c.json').read().strip().splitlines()[-1]); print('$c', d['value'], d['ms_per_step'])"
done
