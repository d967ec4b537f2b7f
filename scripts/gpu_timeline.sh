#!/bin/bash
export TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/timeline; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --steps 40 --warmup 6 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.log)
python scripts/_dbg/iter_timeline.py $OUT/prof | tee $OUT/timeline.txt
find $OUT/prof -name "*.csv" -size +4M -delete
