#!/usr/bin/env python
"""MFMA utilisation per kernel from the rocprofv3 --pmc passes of scripts/pmc_mfma.sh.

  python scripts/pmc_mfma_summary.py gpurun_out/pmc profiles/r02_pmc_mfma_C3.json

Per kernel: sums over launches of SQ_VALU_MFMA_BUSY_CYCLES (cycles a SIMD's matrix pipe is busy, summed over
SIMDs; measured here: 64 per v_mfma_f64_16x16x4_f64), SQ_BUSY_CYCLES (cycles an SQ has work, summed over the 32
shader engines), SQ_INSTS_VALU_MFMA_MOPS_F64 (FP64 matrix operations in units of 512 flops) and GRBM_GUI_ACTIVE
(busy cycles, reported as the SUM over the 8 XCDs: kernel duration in shader cycles = GRBM_GUI_ACTIVE / 8).
mfma_util = MFMA busy cycles / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): the fraction of the chip's matrix-pipe cycles
that were occupied while the kernel ran; flops_from_mops cross-checks the executed flops bench.py prices.
A second argument that is an existing JSON of this script re-derives the summary from its stored counters.
`summary` groups the factorisation's kernels (k_chol_*) and the cluster kernel.
"""
import collections
import csv
import glob
import json
import sys

root, out = sys.argv[1], sys.argv[2]
NUM_SIMD = 256 * 4
NUM_XCD = 8
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(int))
import os
if os.path.isfile(root):  # re-summarise a stored result
    for k, v in json.load(open(root))["kernels"].items():
        for a, b in v["counters"].items():
            agg[k][a] = b
            launches[k][a] = v["launches"]
for f in ([] if os.path.isfile(root) else glob.glob(f"{root}/mfma_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mavba::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k][r["Counter_Name"]] += 1
kernels = {}
for k, c in agg.items():
    n = max(launches[k].values())
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    d = dict(launches=n, counters={a: b for a, b in c.items()})
    if gui > 0:
        d["mfma_util"] = busy / (gui / NUM_XCD * NUM_SIMD)
    if "SQ_INSTS_VALU_MFMA_MOPS_F64" in c:
        d["flops_from_mops_per_launch"] = c["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0 / n
    kernels[k] = d


def group(pred):
    ks = [k for k in kernels if pred(k)]
    gui = sum(kernels[k]["counters"].get("GRBM_GUI_ACTIVE", 0.0) for k in ks)
    busy = sum(kernels[k]["counters"].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for k in ks)
    mops = sum(kernels[k]["counters"].get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) for k in ks)
    return dict(kernels=sorted(ks), mfma_busy_cycles=busy, gui_active_cycles=gui,
                mfma_util=(busy / (gui / NUM_XCD * NUM_SIMD) if gui else None), mfma_mops_f64=mops,
                mfma_flops=mops * 512.0, cycles_per_mfma=(busy / (mops / 4.0) if mops else None))


summary = dict(reduced_solve=group(lambda k: k.startswith("k_chol_")), schur_clusters=group(lambda k: k.startswith("k_schur_clusters")),
               schur_fused=group(lambda k: k.startswith("k_schur_fused") or k.startswith("k_schur_rows")))
json.dump(dict(note=__doc__, summary=summary, kernels=kernels), open(out, "w"), indent=1)
for name, g in summary.items():
    print(name, "mfma_util", g["mfma_util"], "busy", g["mfma_busy_cycles"], "gui", g["gui_active_cycles"])
