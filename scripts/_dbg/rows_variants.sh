#!/bin/bash
# Timing-only builds of k_schur_rows with parts left out (MAVBA_ROWS_SKIP bit mask; results are WRONG - never run tests on them):
#   bash scripts/_dbg/rows_variants.sh 0 1 2 4 ...   ->  scripts/_dbg/_build/libmavba_skip<N>.so
set -e
cd "$(dirname "$0")/../.."
python -m mavmap_amd.build > /dev/null
OBJ=mavmap_amd/lib/obj; mkdir -p scripts/_dbg/_build
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -DMAVBA_ROWS_SKIP=$v ${ROWS_EXTRA:-} -c mavmap_amd/csrc/schur_rows.hip -o /tmp/schur_rows_skip$v.o 2>/dev/null
  objs=$(ls $OBJ/*.o | grep -v schur_rows)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/_dbg/_build/libmavba_skip$v${ROWS_TAG:-}.so $objs /tmp/schur_rows_skip$v.o
  echo built skip $v
done
