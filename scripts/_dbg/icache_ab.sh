#!/bin/bash
# Instruction-cache counters of k_chol_persist: the product build against -DMAVBA_TILE_LA=1 (the look-ahead tile everywhere:
# a smaller kernel). Does the chain pay for the systolic tile's code size?   (own runs, kernel-trace only)
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
R=$PWD
python -m mavmap_amd.build > /dev/null
run() {
  d=$R/gpurun_out/icache_$1; rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d $d -o pmc -- python $R/bench.py --steps 20 --warmup 4 --no-cpu-baseline > /dev/null 2>&1)
  python - "$d" "$1" <<'PY'
import csv, glob, sys, collections
d, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_chol_persist" in k or "k_schur_rows" in k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, v in acc.items():
    calls = max(n[(k, "SQC_ICACHE_REQ")], 1)
    print(tag, k[:40], "per launch:", {c: round(x / calls) for c, x in v.items()})
PY
}
run product
cp mavmap_amd/lib/libmavba.so /tmp/libmavba_keep.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -mllvm -amdgpu-mfma-vgpr-form=1 -DMAVBA_TILE_LA=1 -c mavmap_amd/csrc/dense_chol.hip -o /tmp/dense_chol_la.o 2>/dev/null
objs=$(ls mavmap_amd/lib/obj/*.o | grep -v dense_chol)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mavmap_amd/lib/libmavba.so $objs /tmp/dense_chol_la.o
run tile_la
for c in C3 C2; do timeout 300 python bench.py --config $c --steps 60 --warmup 6 --no-cpu-baseline 2>/tmp/b.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tile_la', d['config']['workload'][:3], d['value'], d['ms_per_step'])"; grep chol_factor /tmp/b.log | head -1; done
cp /tmp/libmavba_keep.so mavmap_amd/lib/libmavba.so
for c in C3 C2; do timeout 300 python bench.py --config $c --steps 60 --warmup 6 --no-cpu-baseline 2>/tmp/b.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('product', d['config']['workload'][:3], d['value'], d['ms_per_step'])"; grep chol_factor /tmp/b.log | head -1; done
