"""Average cycles per phase of k_point_front from a MAVBA_FRONT_TRACE file (debug harness)."""
import sys
import numpy as np
rows = [list(map(int, l.split())) for l in open(sys.argv[1]) if len(l.split()) >= 11]
a = np.array([r[2:11] for r in rows if len(r) >= 11], dtype=np.int64)
wv = np.array([r[1] for r in rows if len(r) >= 11])
names = ["bounds", "jacobian+products", "round0", "round1", "owner", "intr records", "pose compute", "staging+stores"]
print("work-group waves traced:", len(a))
for w in range(4):
    d = np.diff(a[wv == w], axis=1)
    print(f"wave {w}: total {d.sum(1).mean():9.0f} cycles  " + "  ".join(f"{n} {x:7.0f}" for n, x in zip(names, d.mean(0))))
t0, t1 = a[:, 0].min(), a[:, -1].max()
print("kernel span (cycles):", t1 - t0, " work-groups:", len(a) // 4, " per-tile mean:", np.diff(a[wv == 0][:, [0, -1]], axis=1).mean())
