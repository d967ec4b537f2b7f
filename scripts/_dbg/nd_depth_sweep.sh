for d in 0 1 2 3 4; do echo "ND_DEPTH=$d"; MAVBA_ND_DEPTH=$d timeout 200 python bench.py --config C2 --steps 30 --warmup 4 --no-cpu-baseline 2>&1 | grep -E "chol_factor|chol_backsolve|^\{" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  value',d['value'], 'reduced', {k:d['reduced_system'].get(k) for k in ('n','fronts','chain_steps','tiles','nd_parts')})
    else: print(l.rstrip())
"; done
