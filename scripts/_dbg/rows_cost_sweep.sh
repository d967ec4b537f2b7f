export TMPDIR=/tmp
for cost in "20000,15400,17500,27500" "12000,15400,17500,27500" "30000,15400,17500,27500" "20000,15400,19500,30000" "12000,15400,19500,30000" "8000,15400,18500,30000"; do
  echo "ROWS_COST $cost: $(MAVBA_ROWS_COST=$cost MAVBA_CLUSTER_STATS=1 timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline 2>/tmp/b.log | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={r["kernel"]:r["avg_ms"] for r in d["kernels"]}; print(d["value"], d["ms_per_step"], "rows", k.get("schur_fused"), "finalize", k.get("schur_finalize"))') $(grep -h "class [012]" /tmp/b.log | sort | uniq | awk '{print $7"cl/"$11"b"}' | tr '\n' ' ')"
done
