#!/bin/bash
# A/B/A/B of the product library with and without one environment variable, in one visit:  ab_env.sh VAR=VALUE ["C3 C2"]
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
KV=$1; CFGS=${2:-"C3 C2"}
bench() { for c in $CFGS; do timeout 300 python bench.py --config $c --steps 60 --warmup 6 --no-cpu-baseline 2>/tmp/b.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={x['kernel']:x['avg_ms'] for x in d['kernels']}
print('$1', d['config']['workload'][:3], d['value'], d['ms_per_step'], 'factor', k.get('chol_factor'), 'backsolve', k.get('chol_backsolve'))"; done; }
for r in 1 2; do bench base; export $KV; bench $KV; unset ${KV%%=*}; done
