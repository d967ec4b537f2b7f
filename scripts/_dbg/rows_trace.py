"""Average cycles per phase of k_schur_rows from a MAVBA_ROWS_TRACE file (debug harness; class-0 clusters, second batch).

  MAVBA_ROWS_TRACE=/tmp/r.txt python bench.py --steps 8 --warmup 2 --no-cpu-baseline; python scripts/_dbg/rows_trace.py /tmp/r.txt
"""
import sys
import numpy as np
rows = [list(map(int, l.split())) for l in open(sys.argv[1])]
n = max(len(r) for r in rows)
rows = [r for r in rows if len(r) == n]
a = np.array([r[2:] for r in rows], dtype=np.int64)
wv = np.array([r[1] for r in rows])
names = ["jacobian", "wk+sums+factors", "barrier+entries+barrier", "loads+mfma", "next-top"]
print("clusters traced:", len(a) // 4, "(second batch of each), stamps per wave:", a.shape[1])
for w in range(4):
    d = np.diff(a[wv == w], axis=1)
    print(f"wave {w}: total {d.sum(1).mean():8.0f} ticks  " + "  ".join(f"{nm} {x:7.0f}" for nm, x in zip(names, d.mean(0))))
