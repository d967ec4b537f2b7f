export TMPDIR=/tmp
for cfg in C3 C2; do
for cost in "6000,15400,17500,27500" "20000,15400,17500,27500" "20000,13000,15000,24000" "10000,13000,17000,30000" "3000,13000,14000,24000"; do echo "$cfg COST $cost: $(MAVBA_ROWS_COST=$cost python scripts/_dbg/time_front.py $cfg 2>&1 | tail -1)"; done
for pts in 32 64 128 256; do echo "$cfg POINTS $pts: $(MAVBA_CLUSTER_POINTS=$pts python scripts/_dbg/time_front.py $cfg 2>&1 | tail -1)"; done
done
