#!/bin/bash
# A/B in one visit: the product library against a build WITH the in-launch back-substitution tail compiled in (-DMAVBA_BS_IN_LAUNCH;
# round 5: its presence alone cost the forward pass 1-2 us, switched on - MAVBA_CHOL_BACKSOLVE_IN_LAUNCH=1 - it is slower).
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
python -m mavmap_amd.build > /dev/null
bench() { for c in C3 C2; do timeout 300 python bench.py --config $c --steps 60 --warmup 6 --no-cpu-baseline 2>/tmp/b.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['config']['workload'][:3], d['value'], d['ms_per_step'])"; grep chol_factor /tmp/b.log | head -1; done; }
bench product; bench product
cp mavmap_amd/lib/libmavba.so /tmp/libmavba_keep.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -mllvm -amdgpu-mfma-vgpr-form=1 -DMAVBA_BS_IN_LAUNCH -c mavmap_amd/csrc/dense_chol.hip -o /tmp/dense_chol_nt.o 2>/dev/null
objs=$(ls mavmap_amd/lib/obj/*.o | grep -v dense_chol)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mavmap_amd/lib/libmavba.so $objs /tmp/dense_chol_nt.o
bench tail_compiled_in; MAVBA_CHOL_BACKSOLVE_IN_LAUNCH=1 bench tail_switched_on
cp /tmp/libmavba_keep.so mavmap_amd/lib/libmavba.so
