#!/bin/bash
# From how many partials per block the pre-reduction launch pays (MAVBA_PRE_REDUCE_FROM): pre_reduce_ab.sh "C2" "64 128 256 4096"
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
for c in ${1:-C2}; do for v in ${2:-64 128 256 4096}; do
  echo "$c PRE_REDUCE_FROM=$v $(MAVBA_PRE_REDUCE_FROM=$v timeout 300 python bench.py --config $c --steps 60 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={r["kernel"]:r["avg_ms"] for r in d["kernels"]}; print(d["value"], d["ms_per_step"], "finalize", k.get("schur_finalize"))')"
done; done
