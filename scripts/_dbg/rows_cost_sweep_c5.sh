#!/bin/bash
# Cluster cost model of k_schur_rows at C5 (per cluster, per batch of the 80 / 96 / 128-row class): bench line per setting
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
for cost in "20000,15400,17500,27500" "40000,15400,17500,27500" "80000,15400,17500,27500" "160000,15400,17500,27500" "40000,15400,17500,22000"; do
  echo "C5 ROWS_COST $cost: $(MAVBA_ROWS_COST=$cost timeout 300 python bench.py --config C5 --steps 30 --warmup 4 --no-cpu-baseline 2>/tmp/b.log | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={r["kernel"]:r["avg_ms"] for r in d["kernels"]}; r=d["reduced_system"]; print(d["value"], d["ms_per_step"], "rows", k.get("schur_fused"), "finalize", k.get("schur_finalize"), "clusters", r["schur_clusters"], "partials", r["cluster_partials"])')"
done
