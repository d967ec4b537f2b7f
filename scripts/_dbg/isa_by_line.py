"""Vector-ALU instructions of one kernel by source line (no GPU needed): where do the issue slots of a kernel go?

  cd /tmp/x && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -gline-tables-only --save-temps \
      -c /root/repo/mavmap_amd/csrc/schur_rows.hip -o x.o
  python scripts/_dbg/isa_by_line.py /tmp/x/schur_rows-hip-amdgcn-amd-amdhsa-gfx950.s k_schur_rowsILi8ELb0ELb0 40

Counts are static (every instantiation of a loop body once, branches not weighed); matrix instructions are listed apart.
Round 5: k_schur_rows<8, false, false> has ~7 000 vector instructions in three instantiations of its batch loop, 2 000 of them
FP64 - the reduce-scatter's DPP moves and selects, the emit's index arithmetic and the zeroing of inactive lanes' values are the
rest (profiles/r05_isa_by_line_k_schur_rows.txt)."""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
files, start, end = {}, None, None
for i, l in enumerate(lines):
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
    if re.match(r"^_ZN5mavba\w*" + pat + r"\w*:", l):
        start = i
    if start is not None and end is None and l.strip().startswith(".amdhsa_kernel"):
        end = i
cur = None
valu, f64, kinds = collections.Counter(), collections.Counter(), collections.Counter()
mfma = 0
for l in lines[start:end]:
    m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    m = re.match(r"\s+(v_\w+)", l)
    if not m:
        continue
    op = m.group(1)
    if op.startswith("v_mfma"):
        mfma += 1
        continue
    valu[cur] += 1
    kinds[re.sub(r"_e32$|_e64$|_dpp$|_sdwa$", "", op)] += 1
    if "f64" in op:
        f64[cur] += 1
print("vector instructions %d, of them FP64 %d; matrix instructions %d" % (sum(valu.values()), sum(f64.values()), mfma))
print("by opcode:", ", ".join("%s %d" % kv for kv in kinds.most_common(14)))
for k, v in valu.most_common(top):
    print("%-22s:%-5d  vector %5d   fp64 %5d   other %5d" % (k[0], k[1], v, f64[k], v - f64[k]))
