#!/bin/bash
# Points per cluster of k_schur_rows (MAVBA_CLUSTER_POINTS) against the bench line:  cluster_points_sweep.sh "C2" "0 32 48 64 96 128"   (0 = default rule)
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
for c in ${1:-C2}; do for n in ${2:-0 32 48 64 96 128}; do
  if [ "$n" = 0 ]; then unset MAVBA_CLUSTER_POINTS; else export MAVBA_CLUSTER_POINTS=$n; fi
  echo "$c CLUSTER_POINTS=$n $(timeout 300 python bench.py --config $c --steps 60 --warmup 6 --no-cpu-baseline 2>/tmp/b.log | python -c '
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={x["kernel"]:x["avg_ms"] for x in d["kernels"]}; r=d["reduced_system"]
print(d["value"], d["ms_per_step"], "rows", k.get("schur_fused"), "finalize", k.get("schur_finalize"), "clusters", r["schur_clusters"], "partials", r["cluster_partials"])')"
done; done
