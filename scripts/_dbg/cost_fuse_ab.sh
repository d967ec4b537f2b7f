# A/B of the candidate cost inside the back-substitution kernel (MAVBA_COST_FUSE_MAX_OBS) at C2
export TMPDIR=/tmp
for v in 200000 400000 200000 400000; do
  echo "C2 COST_FUSE_MAX_OBS=$v $(MAVBA_COST_FUSE_MAX_OBS=$v timeout 300 python bench.py --config C2 --steps 60 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={r["kernel"]:r["avg_ms"] for r in d["kernels"]}; print(d["value"], d["ms_per_step"], "backsub", k.get("backsub_points"), "cost", k.get("cost_only"))')"
done
