export TMPDIR=/tmp
for cfg in C3 C5; do
for cost in "6000,15400,17500,27500" "20000,15400,17500,27500" "40000,15400,17500,27500" "20000,15400,16500,24000" "40000,15400,16000,22000"; do
  MAVBA_ROWS_COST=$cost timeout 300 python bench.py --config $cfg --steps 40 --warmup 6 --no-cpu-baseline 2>/tmp/b.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={r['kernel']:r['avg_ms'] for r in d['kernels']}; print('$cfg', '$cost', d['value'], d['ms_per_step'], 'fused', k.get('schur_fused'), 'finalize', k.get('schur_finalize'), 'clusters', d['reduced_system']['schur_clusters'], 'partials', d['reduced_system']['cluster_partials'], 'setup', d['solve']['setup_seconds'])"
done; done
MAVBA_CLUSTER_POINTS=256 timeout 300 python bench.py --config C3 --steps 40 --warmup 6 --no-cpu-baseline 2>/tmp/b.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={r['kernel']:r['avg_ms'] for r in d['kernels']}; print('C3 pts256', d['value'], d['ms_per_step'], 'fused', k.get('schur_fused'), 'finalize', k.get('schur_finalize'), 'clusters', d['reduced_system']['schur_clusters'], 'partials', d['reduced_system']['cluster_partials'])"
