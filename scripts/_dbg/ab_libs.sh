#!/bin/bash
# A/B/A/B of two builds of the library in one visit:  ab_libs.sh libA.so libB.so ["C3 C2"]   (each is copied over the built one in turn)
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
A=$1; B=$2; CFGS=${3:-"C3 C2"}
LIB=mavmap_amd/lib/libmavba.so; cp $LIB /tmp/libmavba_keep.so
bench() { for c in $CFGS; do timeout 300 python bench.py --config $c --steps 60 --warmup 6 --no-cpu-baseline 2>/tmp/b.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={x['kernel']:x['avg_ms'] for x in d['kernels']}
print('$1', d['config']['workload'][:3], d['value'], d['ms_per_step'], 'rows', k.get('schur_fused'), 'factor', k.get('chol_factor'), 'backsolve', k.get('chol_backsolve'))"; done; }
for r in 1 2; do cp $A $LIB; MAVBA_SKIP_STAMP=1 bench $(basename $A); cp $B $LIB; MAVBA_SKIP_STAMP=1 bench $(basename $B); done
cp /tmp/libmavba_keep.so $LIB
