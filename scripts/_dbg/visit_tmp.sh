export TMPDIR=/tmp MAVBA_SKIP_HEAVY=1
OUT=$PWD/gpurun_out/r04r; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -x -q -p no:cacheprovider --timeout 600 > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 200 python scripts/_dbg/local_ba_latency.py 2>&1 | head -22
for c in C3 C2; do timeout 300 python bench.py --config $c --steps 60 --warmup 6 --no-cpu-baseline > $OUT/bench_$c.json 2> $OUT/bench_$c.log; python -c "
import json; d=json.loads(open('$OUT/bench_$c.json').read().strip().splitlines()[-1]); print('$c', d['value'], d['ms_per_step'], d['ms_per_step_with_event_timers'])"; grep lm_snapshot $OUT/bench_$c.log; done
