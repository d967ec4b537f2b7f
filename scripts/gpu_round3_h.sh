#!/bin/bash
# GPU visit H: device-resident scene route + set-up timing.
set -u
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r03h
mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.build(); print('build ok')" > $OUT/build.log 2>&1; tail -2 $OUT/build.log
timeout 600 python -m pytest tests/test_scene.py tests/test_gpu_setup.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 > $OUT/pytest_scene.log 2>&1
echo "scene tests exit $?"; tail -30 $OUT/pytest_scene.log
timeout 600 python scripts/_dbg/scene_timing.py C3 > $OUT/scene_timing.log 2>&1; grep "route\|flatten\|bundle" $OUT/scene_timing.log | tail -12
MAVBA_SETUP=device timeout 300 python scripts/_dbg/setup_timing.py C3 > $OUT/setup_device.log 2>&1; grep -A22 "mavba_solve call 2" $OUT/setup_device.log | head -30
