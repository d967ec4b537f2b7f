"""Kernel-trace CSV of rocprofv3 -> busy time / gaps of the steady-state part (last 60 % of the kernels)."""
import csv, sys, glob, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
rows = rows[int(0.4 * n):]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
print("kernels %d span %.3f ms busy %.3f ms (%.0f %%) median gap %.2f us mean gap %.2f us" % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, sorted(gaps)[len(gaps) // 2] / 1e3, sum(gaps) / len(gaps) / 1e3))
per = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:60]; per[k][0] += 1; per[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:25]:
    print("  %-62s n=%5d avg %7.2f us" % (k, c, t / c / 1e3))
