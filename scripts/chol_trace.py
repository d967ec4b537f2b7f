#!/usr/bin/env python
"""Where does the persistent factorisation's chain spend its time?  (profiling aid)

  MAVBA_CHOL_TRACE=/tmp/trace.txt python scripts/chol_trace.py [C3|C2|C5] [scale]

Runs a few LM iterations so that the last solve's stamps (100 MHz wall clock) are dumped when the session closes,
then prints per chain column: wait for the helpers' two tiles (+ their loads when they were not prefetched), the first 32
rows of the panel solve, the tile factor + inverse (which now contains the rest of the panel solve and the diagonal update),
publish; and how late the helpers' PRE tasks were."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = os.environ.setdefault("MAVBA_CHOL_TRACE", "/tmp/chol_trace.txt")
import numpy as np
import mavmap_amd
from mavmap_amd import synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
p = synth.make_config(cfg, scale=scale)
with mavmap_amd.Session(p, dict(max_num_iterations=200, function_tolerance=1e-6, gradient_tolerance=1e-10)) as s:
    s.iterate(4)
C, T = [], []
for line in open(path):
    f = line.split()
    if f[0] == "C":
        C.append([int(x) for x in f[1:]])
    elif f[0] == "T":
        T.append([int(x) for x in f[1:]])
    else:
        print(line.strip())
C = np.array(C, dtype=np.int64)
T = np.array(T, dtype=np.int64)
t0 = min(C[:, 2][C[:, 2] > 0].min(), T[:, 5][T[:, 5] > 0].min())
us = lambda x: (x - t0) / 100.0
print("col node |  start  wait+load  trsm(rows 0-31)  factor(+rest of trsm, update)  publish | end   (us; durations)")
for r in C:
    j, n, t = r[0], r[1], r[2:]
    if t[0] == 0:
        continue
    sub = t[1] > 0
    seq = [t[0], t[1], t[2], t[6], t[7]]
    d = np.diff(seq) / 100.0
    print(f"{j:3d} {n:3d} | {us(t[0]):7.1f} " + " ".join(f"{x:7.1f}" for x in d) + f" | {us(t[7]):7.1f}")
ok = C[:, 2] > 0
ticks = (C[ok, 2 + 4] - C[ok, 2 + 3]).astype(float)
wall = (C[ok, 2 + 6] - C[ok, 2 + 2]).astype(float) / 100.0
good = (ticks > 0) & (wall > 0)
if good.any():
    print("shader clock over the tile factorisations (s_memtime ticks per us of the 100 MHz clock): median %.0f, min %.0f, max %.0f"
          % (np.median(ticks[good] / wall[good]), (ticks[good] / wall[good]).min(), (ticks[good] / wall[good]).max()))
print("total forward us:", us(max(C[:, 9].max(), T[:, 8].max())))
# helpers: lateness of PRE tasks relative to when the chain started waiting for them
kinds = {0: "TILE", 1: "PRE_DIAG", 2: "PRE_SUB"}
for kind in (1, 2):
    sel = T[T[:, 1] == kind]
    if len(sel):
        dur = (sel[:, 8] - sel[:, 5]) / 100.0
        upd = (sel[:, 6] - sel[:, 5]) / 100.0
        print(f"{kinds[kind]}: {len(sel)} tasks, mean duration {dur.mean():.1f} us (updates incl. waits {upd.mean():.1f}), mean #updates {sel[:, 4].mean():.1f}")
sel = T[T[:, 1] == 0]
print(f"TILE: {len(sel)} tasks; updates phase {((sel[:, 6] - sel[:, 5]) / 100.0).mean():.1f} us, wait dflag {((sel[:, 7] - sel[:, 6]) / 100.0).mean():.1f} us, "
      f"solve+publish {((sel[:, 8] - sel[:, 7]) / 100.0).mean():.1f} us")
