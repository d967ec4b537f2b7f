export TMPDIR=/tmp MAVBA_SKIP_HEAVY=1
OUT=$PWD/gpurun_out/r04q; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider --timeout 600 --deselect tests/test_gpu_fullsize.py > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
timeout 200 python scripts/_dbg/local_ba_latency.py 2>&1 | head -3
for c in C3 C2; do timeout 300 python bench.py --config $c --steps 60 --warmup 6 --no-cpu-baseline > $OUT/bench_$c.json 2> $OUT/bench_$c.log; python -c "
import json; d=json.loads(open('$OUT/bench_$c.json').read().strip().splitlines()[-1]); print('$c', d['value'], d['ms_per_step'], d['ms_per_step_with_event_timers'])"; done
MAVBA_SPECULATE=0 timeout 300 python bench.py --config C3 --steps 60 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3 nospec', d['value'], d['ms_per_step'])"
MAVBA_SPECULATE=0 timeout 200 python scripts/_dbg/local_ba_latency.py 2>&1 | head -3
