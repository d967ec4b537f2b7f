"""Time line of one k_schur_rows launch from a MAVBA_ROWS_TRACE file (debug harness, round 6): per cluster the ticks of
its tables (entry -> first batch), its batches and its emit (differences of stamps inside one work-group).

  MAVBA_ROWS_TRACE=/tmp/r.txt python bench.py --steps 8 --warmup 2 --no-cpu-baseline; python scripts/_dbg/rows_timeline.py /tmp/r.txt
"""
import sys
import collections
import numpy as np
rows = [list(map(int, l.split())) for l in open(sys.argv[1])]
a = np.array([r for r in rows if len(r) == 17 and r[1] == 0], dtype=np.int64)  # wave 0 of every cluster
ent, l0, l1, end, hw, meta = a[:, 11], a[:, 12], a[:, 13], a[:, 14], a[:, 15], a[:, 16]
nb, cls = meta >> 8, meta & 255
print("clusters %d, batches %d" % (len(a), nb.sum()))
print("per cluster (mean ticks): tables %.0f, batches %.0f (%.0f per batch), emit+cost %.0f" %
      ((l0 - ent).mean(), (l1 - l0).mean(), (l1 - l0).sum() / nb.sum(), (end - l1).mean()))
for c in sorted(set(cls)):
    m = cls == c
    print("  row class %d: %5d clusters, %6d batches, %.0f ticks per batch, tables %.0f, emit %.0f" %
          (c, m.sum(), nb[m].sum(), (l1 - l0)[m].sum() / nb[m].sum(), (l0 - ent)[m].mean(), (end - l1)[m].mean()))
tot = (end - ent).sum()
print("sum over clusters: tables %.1f %%, batches %.1f %%, emit %.1f %% of the work-group time" %
      (100.0 * (l0 - ent).sum() / tot, 100.0 * (l1 - l0).sum() / tot, 100.0 * (end - l1).sum() / tot))
# (no launch-wide time line: s_memtime stamps of different CUs do not share an origin on this part - only differences inside a work-group are used)
