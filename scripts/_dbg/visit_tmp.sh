export TMPDIR=/tmp MAVBA_SKIP_HEAVY=1
OUT=$PWD/gpurun_out/r04n; mkdir -p $OUT
MAVBA_CHOL_PERSIST=0 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_filter.py -x -q -p no:cacheprovider 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -p no:cacheprovider -k "c5_shaped or c5_full_size_step or dense" 2>&1 | tail -3
timeout 400 python bench.py --config C5 --steps 30 --warmup 4 --no-cpu-baseline > $OUT/bench_C5.json 2> $OUT/bench_C5.log
grep -E "schur_fused|chol_factor|memset_S|point_front|schur_chunks|schur_finalize|backsub|camera_sweep" $OUT/bench_C5.log | head -12
python -c "
import json; d=json.loads(open('$OUT/bench_C5.json').read().strip().splitlines()[-1]); print('C5', d['value'], d['ms_per_step'])"
