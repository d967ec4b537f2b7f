#!/bin/bash
# Depth of the nested dissection of the reduced camera system (MAVBA_ND_DEPTH) against the bench line:  nd_depth_sweep.sh "C3 C2" "2 3 4"
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
for c in ${1:-C3 C2}; do for d in ${2:-2 3 4}; do echo "$c ND_DEPTH=$d $(MAVBA_ND_DEPTH=$d timeout 300 python bench.py --config $c --steps ${STEPS:-60} --warmup 6 --no-cpu-baseline 2>/tmp/b.log | python -c '
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k={x["kernel"]:x["avg_ms"] for x in d["kernels"]}; r=d["reduced_system"]
print(d["value"], d["ms_per_step"], "factor", k.get("chol_factor"), "backsolve", k.get("chol_backsolve"), "dim", r["matrix_dim"], "envelope tiles", r["envelope_tiles"], "fronts", r["nd_parts"], "setup", d["solve"]["setup_seconds"])')"; done; done
