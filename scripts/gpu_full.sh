#!/bin/bash
# Full GPU suite + smoke (the round-end check), logs under gpurun_out/full
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/full; mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.build(); print('build ok')" > $OUT/build.log 2>&1; tail -1 $OUT/build.log
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -6 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $OUT/smoke.log
