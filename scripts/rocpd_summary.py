#!/usr/bin/env python
"""Per-kernel summary (calls, total, average, min, max, share) of a rocprofv3 rocpd .db —
the same numbers `rocprofv3 --stats` prints, written as CSV for profiles/."""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                 "from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
with open(out, "w") as f:
    f.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage\n")
    for r in rows:
        name = r[0].replace('"', "'")
        f.write(f"\"{name}\",{r[1]},{r[2]},{r[3]:.1f},{r[4]},{r[5]},{100.0 * r[2] / tot:.2f}\n")
print(open(out).read())
