"""Time line of one k_schur_rows launch from a MAVBA_ROWS_TRACE file (debug harness, round 6): per cluster the ticks of
its tables (entry -> first batch), its batches and its emit, how the clusters sit on the CUs, and the launch's tail.

  MAVBA_ROWS_TRACE=/tmp/r.txt python bench.py --steps 8 --warmup 2 --no-cpu-baseline; python scripts/_dbg/rows_timeline.py /tmp/r.txt
"""
import sys
import collections
import numpy as np
rows = [list(map(int, l.split())) for l in open(sys.argv[1])]
a = np.array([r for r in rows if len(r) == 17 and r[1] == 0], dtype=np.int64)  # wave 0 of every cluster
ent, l0, l1, end, hw, meta = a[:, 11], a[:, 12], a[:, 13], a[:, 14], a[:, 15], a[:, 16]
nb, cls = meta >> 8, meta & 255
# (s_memtime is a counter per XCD: put every XCD's first entry at 0)
xcc = (hw >> 32) & 15
for x in set(xcc.tolist()):
    m = xcc == x
    base = ent[m].min()
    for v in (ent, l0, l1, end):
        v[m] -= base
t0, t1 = ent.min(), end.max()
print("clusters %d, batches %d; launch span %d ticks (XCDs end at %s)" % (len(a), nb.sum(), t1 - t0, " ".join(str(int(end[xcc == x].max())) for x in sorted(set(xcc.tolist())))))
print("per cluster (mean ticks): tables %.0f, batches %.0f (%.0f per batch), emit+cost %.0f" %
      ((l0 - ent).mean(), (l1 - l0).mean(), (l1 - l0).sum() / nb.sum(), (end - l1).mean()))
for c in sorted(set(cls)):
    m = cls == c
    print("  row class %d: %5d clusters, %6d batches, %.0f ticks per batch, tables %.0f, emit %.0f" %
          (c, m.sum(), nb[m].sum(), (l1 - l0)[m].sum() / nb[m].sum(), (l0 - ent)[m].mean(), (end - l1)[m].mean()))
tot = (end - ent).sum()
print("sum over clusters: tables %.1f %%, batches %.1f %%, emit %.1f %% of the work-group time" %
      (100.0 * (l0 - ent).sum() / tot, 100.0 * (l1 - l0).sum() / tot, 100.0 * (end - l1).sum() / tot))
# where the clusters ran: XCC, SE, SH, CU from HW_ID (cu_id bits 11:8, sh_id 12, se_id 15:13) and XCC_ID (bits 3:0 of the upper word)
cu = ((hw >> 32) & 15) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 15)
per = collections.defaultdict(list)
for k in range(len(a)):
    per[int(cu[k])].append((int(ent[k]), int(end[k])))
busy, last, first = [], [], []
for k, iv in per.items():
    iv.sort()
    # union length of the intervals (two work-groups share a CU)
    u, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            u += cur_e - cur_s; cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    u += cur_e - cur_s
    busy.append(u); last.append(max(e for _, e in iv)); first.append(iv[0][0])
busy, last, first = np.array(busy), np.array(last), np.array(first)
print("CUs seen %d; a CU has a work-group resident %.1f %% of the span on average (min %.1f %%); first entry spread %d ticks; "
      "CUs finish %d .. %d ticks before the end (mean %d)" %
      (len(per), 100.0 * busy.mean() / (t1 - t0), 100.0 * busy.min() / (t1 - t0), first.max() - t0, (t1 - last).min(), (t1 - last).max(), (t1 - last).mean()))
occ = sum((end - ent)) / float((t1 - t0) * len(per))
print("average resident work-groups per CU over the span: %.2f" % occ)
