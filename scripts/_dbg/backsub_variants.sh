#!/bin/bash
# Timing-only builds of k_backsub_points_packed (MAVBA_BS_SKIP: 1 = no per-observation arithmetic, 2 = owner lanes add ONE
# observation instead of all of the point's; results are WRONG): where does the kernel's time go?
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
python -m mavmap_amd.build > /dev/null
cp mavmap_amd/lib/libmavba.so /tmp/libmavba_keep.so
objs=$(ls mavmap_amd/lib/obj/*.o | grep -v kernels.o)
for v in 0 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -DMAVBA_BS_SKIP=$v -c mavmap_amd/csrc/kernels.hip -o /tmp/kernels_v$v.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mavmap_amd/lib/libmavba.so $objs /tmp/kernels_v$v.o
  for c in C3 C2; do timeout 300 python bench.py --config $c --steps 40 --warmup 6 --no-cpu-baseline 2>/tmp/b.log >/dev/null; echo "skip $v $c $(grep backsub_points /tmp/b.log | head -1)"; done
done
cp /tmp/libmavba_keep.so mavmap_amd/lib/libmavba.so
