"""Average cycles per phase of k_schur_rows from a MAVBA_ROWS_TRACE file (debug harness; second batch of the clusters of one row class,
default class 1 = 96 rows).

  MAVBA_ROWS_TRACE=/tmp/r.txt python bench.py --steps 8 --warmup 2 --no-cpu-baseline; python scripts/_dbg/rows_trace.py /tmp/r.txt
"""
import sys
import numpy as np
# a line: cluster, wave, number of phase stamps, 8 phase stamps, then the cluster's time line (scripts/_dbg/rows_timeline.py reads that)
rows = [list(map(int, l.split())) for l in open(sys.argv[1])]
rows = [r for r in rows if len(r) == 17 and r[2] == 5 and (r[16] & 255) == int(sys.argv[2] if len(sys.argv) > 2 else 1)]  # second batch of a cluster of the chosen row class
a = np.array([r[3:8] for r in rows], dtype=np.int64)
wv = np.array([r[1] for r in rows])
names = ["jacobian", "sums+factors", "entries+barrier", "loads+mfma"]
print("clusters traced:", len(a) // 4, "(second batch of each), stamps per wave:", a.shape[1])
for w in range(4):
    d = np.diff(a[wv == w], axis=1)
    print(f"wave {w}: total {d.sum(1).mean():8.0f} ticks  " + "  ".join(f"{nm} {x:7.0f}" for nm, x in zip(names, d.mean(0))))
