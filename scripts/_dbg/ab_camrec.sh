#!/bin/bash
# A/B in one visit: camera records as matrices (R, t, Jl: 21 values) against the round-4 records (w, t, a, b, c) built from the
# previous commit (mavmap_amd/lib/libmavba_base.so, made by hand), and the Jacobian probe at one / two work-groups per CU.
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
L=mavmap_amd/lib
cp $L/libmavba.so /tmp/libmavba_keep.so
bench() { for c in $2; do timeout 300 python bench.py --config $c --steps 60 --warmup 6 --no-cpu-baseline 2>/tmp/b.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={x['kernel']:x['avg_ms'] for x in d['kernels']}
print('$1', d['config']['workload'][:3], d['value'], d['ms_per_step'], 'rows', k.get('schur_fused'), 'sweep', k.get('camera_sweep'), 'backsub', k.get('backsub_points'), 'cost', k.get('cost_only'), 'probe', d['jacobian_sweep'].get('frac'), d['jacobian_sweep'].get('avg_ms'))"; done; }
for v in base new base new; do
  if [ $v = base ]; then cp $L/libmavba_base.so $L/libmavba.so; else cp /tmp/libmavba_keep.so $L/libmavba.so; fi
  bench $v "C3 C2"
done
cp $L/libmavba_w1.so $L/libmavba.so; bench probe_one_group C3
cp /tmp/libmavba_keep.so $L/libmavba.so
