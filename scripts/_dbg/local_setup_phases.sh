mkdir -p gpurun_out/lw
MAVBA_SETUP_TIMING=1 timeout 200 python scripts/_dbg/local_ba_latency.py > gpurun_out/lw/phases.log 2>&1
grep -n "call 20" -A45 gpurun_out/lw/phases.log | head -60
