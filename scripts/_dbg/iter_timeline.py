"""One LM iteration's kernel timeline (start offset, duration, gap before) from a rocprofv3 kernel-trace CSV (debug harness).

  python scripts/_dbg/iter_timeline.py <dir with *kernel_trace.csv> [first]
("first": the first half of the iterations - bench.py's un-instrumented pass; its second pass carries HIP events.)
Prints the median over the steady-state iterations, an iteration = the kernels from one k_schur_fused / k_point_front to the next."""
import csv, sys, glob, statistics
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mavba::", "")[:44]
starts = [i for i, r in enumerate(rows) if name(r).startswith("k_schur_fused") or name(r).startswith("k_schur_rows") or name(r).startswith("k_point_front<8, true")]
iters = [rows[a:b] for a, b in zip(starts, starts[1:])]
iters = iters[len(iters) // 8:len(iters) // 2 - 2] if len(sys.argv) > 2 and sys.argv[2] == "first" else iters[len(iters) // 3:]
seq = statistics.mode(tuple(name(r) for r in it) for it in iters)  # (the most frequent kernel sequence = a steady-state iteration)
L = len(seq)
iters = [it for it in iters if tuple(name(r) for r in it) == seq]
print("iterations used %d, kernels per iteration %d" % (len(iters), L))
tot_gap = 0.0
for k in range(L):
    dur = statistics.median(int(it[k]["End_Timestamp"]) - int(it[k]["Start_Timestamp"]) for it in iters) / 1e3
    gap = statistics.median(int(it[k]["Start_Timestamp"]) - int(it[k - 1]["End_Timestamp"]) for it in iters) / 1e3 if k else 0.0
    tot_gap += gap
    print("  %-46s gap %6.2f us  run %8.2f us" % (name(iters[0][k]), gap, dur))
period = statistics.median(int(b[0]["Start_Timestamp"]) - int(a[0]["Start_Timestamp"]) for a, b in zip(iters, iters[1:]) if True) / 1e3
last_gap = statistics.median(int(b[0]["Start_Timestamp"]) - int(a[-1]["End_Timestamp"]) for a, b in zip(iters, iters[1:])) / 1e3
print("period %.1f us; gaps inside %.1f us; gap to the next iteration %.1f us" % (period, tot_gap, last_gap))
