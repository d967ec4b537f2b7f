# A/B of two library builds on one box: local-window latency and the C3 bench line (debug harness)
for r in 1 2; do for v in A B; do cp mavmap_amd/lib/libmavba_$v.so mavmap_amd/lib/libmavba.so; echo "variant $v: $(timeout 200 python scripts/_dbg/local_ba_latency.py 2>&1 | sed -n 2p) | C3 $(timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"; done; done
cp mavmap_amd/lib/libmavba_B.so mavmap_amd/lib/libmavba.so
