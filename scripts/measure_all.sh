#!/bin/bash
# One GPU-box visit for the round's committed evidence: benches, rocprofv3 kernel stats, PMC passes.
# Usage (repo root on the GPU box): bash scripts/measure_all.sh <round-tag>
set -u
export TMPDIR=/tmp
TAG=${1:-r02}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
nproc > $OUT/nproc.txt
# kernel trace + stats (own run)
rm -rf $OUT/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --steps 60 --warmup 6 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof_bench.log); echo "rocprof rc $?"
find $OUT/prof -name "*kernel_stats*.csv" -exec cp {} $OUT/bench_C3_kernel_stats.csv \;
find $OUT/prof -name "*kernel_trace*.csv" -size +8M -delete
# PMC passes (each its own run, kernel-trace only)
rm -rf $R/gpurun_out/pmc
bash scripts/pmc_traffic.sh > $OUT/pmc_traffic.log 2>&1
python scripts/pmc_summary.py $R/gpurun_out/pmc $OUT/pmc_traffic_C3.json > $OUT/pmc_traffic_summary.txt 2>&1
bash scripts/pmc_mfma.sh > $OUT/pmc_mfma.log 2>&1
python scripts/pmc_mfma_summary.py $R/gpurun_out/pmc $OUT/pmc_mfma_C3.json > $OUT/pmc_mfma_summary.txt 2>&1
cp $R/gpurun_out/pmc/mfma_counter_names.txt $OUT/ 2>/dev/null
rm -rf $R/gpurun_out/pmc $OUT/prof
# the bench lines last: they quote the counter passes above (copied to profiles/ on this box, and again by the caller)
cp $OUT/pmc_traffic_C3.json $R/profiles/${TAG}_pmc_traffic_C3.json
cp $OUT/pmc_mfma_C3.json $R/profiles/${TAG}_pmc_mfma_C3.json
for C in C3 C2 C5; do
  timeout 600 python bench.py --config $C --steps 60 --warmup 6 > $OUT/bench_$C.json 2> $OUT/bench_$C.log; echo "bench $C rc $?"
done
ls -la $OUT
