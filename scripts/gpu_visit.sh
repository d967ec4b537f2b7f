#!/bin/bash
# One GPU-box visit while tuning: [tests <pytest args>] then bench lines of the named configs.
#   bash scripts/gpu_visit.sh <tag> "<pytest args or empty>" "<configs, e.g. C3 C2>"
set -u
export TMPDIR=/tmp
TAG=${1:-visit}; PYT=${2:-}; CFGS=${3:-C3 C2}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
if [ -n "$PYT" ]; then
  timeout 1500 python -m pytest $PYT -q --tb=short -p no:cacheprovider --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -25 $OUT/pytest.log
fi
for c in $CFGS; do
  MAVBA_CLUSTER_STATS=1 timeout 400 python bench.py --config $c --steps 60 --warmup 6 --no-cpu-baseline > $OUT/bench_$c.json 2> $OUT/bench_$c.log
  grep -E "cluster stats" $OUT/bench_$c.log | sort | uniq | head -4
  grep -E "^\| .(schur_fused|chol_factor|chol_backsolve|backsub_points|camera_sweep|schur_finalize|point_front)" $OUT/bench_$c.log | head -8
  python -c "
import json; d=json.loads(open('$OUT/bench_$c.json').read().strip().splitlines()[-1]); print('$c', d['value'], d['ms_per_step'])"
done
