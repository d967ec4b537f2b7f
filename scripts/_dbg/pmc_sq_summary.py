"""Per-kernel averages of the SQ counters collected by scripts/pmc.sh (debug aid): python scripts/_dbg/pmc_sq_summary.py gpurun_out/pmc <tag>"""
import collections, csv, glob, sys
root, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mavba::", "")[:34]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
keys = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS",
        "SQ_INSTS_VALU", "SQ_INSTS_VMEM", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"]
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0])) / max(len(kv[1].get("SQ_WAVE_CYCLES", [1])), 1)):
    wc = sum(d.get("SQ_WAVE_CYCLES", [0])) / max(len(d.get("SQ_WAVE_CYCLES", [1])), 1)
    if wc <= 0:
        continue
    print(f"{k:34s}", " ".join(f"{n.replace('SQ_', '')}={sum(d[n]) / len(d[n]) / (wc if n not in ('SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'GRBM_GUI_ACTIVE') and not n.startswith('SQ_INSTS') else 1):.3g}" for n in keys if n in d))
