#!/bin/bash
# The camera sweep as extra work-groups of the k_schur_rows launch beyond local windows (MAVBA_SWEEP_RIDE_MAX_OBS): A/B per config
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
for c in ${1:-C2 C3}; do for v in 0 100000000 0 100000000; do
  echo "$c SWEEP_RIDE_MAX_OBS=$v $(MAVBA_SWEEP_RIDE_MAX_OBS=$v timeout 300 python bench.py --config $c --steps 60 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={r["kernel"]:r["avg_ms"] for r in d["kernels"]}; print(d["value"], d["ms_per_step"], "rows", k.get("schur_fused"), "sweep", k.get("camera_sweep"))')"
done; done
