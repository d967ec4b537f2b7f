"""Average cycles per phase of k_schur_fused from a MAVBA_FUSED_TRACE file (debug harness).

  MAVBA_FUSED_TRACE=/tmp/f.txt python bench.py --steps 8 --warmup 2 --no-cpu-baseline; python scripts/_dbg/fused_trace.py /tmp/f.txt
"""
import sys
import numpy as np
rows = [list(map(int, l.split())) for l in open(sys.argv[1])]
rows = [r for r in rows if len(r) == 9]
a = np.array([r[2:] for r in rows], dtype=np.int64)
wv = np.array([r[1] for r in rows])
names = ["tables+jacobian", "round0", "round1", "owner+clear", "entry matrix", "mfma"]
print("clusters traced:", len(a) // 8, "(second batch of each)")
for w in range(8):
    d = np.diff(a[wv == w], axis=1)
    print(f"wave {w}: total {d.sum(1).mean():8.0f} ticks  " + "  ".join(f"{n} {x:7.0f}" for n, x in zip(names, d.mean(0))))
