#!/usr/bin/env python3
"""Developer tool: per-kernel register summary + instruction-class histogram between barriers of one kernel in a .s file.
usage: isa_summary.py file.s kernel_name_substring"""
import sys, re
from collections import Counter
src = open(sys.argv[1]).read()
pat = sys.argv[2]
for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", src):
    pass
names = re.findall(r"\.name:\s+(\S+)", src)
meta = re.findall(r"\.private_segment_fixed_size:\s+(\d+)\n\s+\.sgpr_count:\s+(\d+)\n\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n){0,6}?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", src)
lines = src.split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l and l.split()[0].endswith(":"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
def cls(op):
    if op.startswith("v_mfma"): return "MFMA"
    if op.startswith("v_accvgpr"): return "ACCMOV"
    if op.startswith(("v_readlane", "v_writelane")): return "SSPILL"
    if op.startswith("v_"): return "VALU"
    if op.startswith("s_"): return "SALU"
    if op.startswith("ds_"): return "LDS"
    return op
bars = [i for i, l in enumerate(body) if l.strip().startswith("s_barrier")]
hdrs = [i for i, l in enumerate(body) if "Loop Header" in l]
print("lines", len(body), "barriers", bars, "n_loop_headers", len(hdrs))
pts = sorted(set([0] + bars + [len(body)]))
for a, b in zip(pts[:-1], pts[1:]):
    c = Counter()
    for l in body[a:b]:
        t = l.strip().split()
        if not t or t[0].startswith((";", ".")) or t[0].endswith(":"): continue
        c[cls(t[0])] += 1
    print(a, b, dict(c))
print("scratch ops:", sum(1 for l in body if "scratch_" in l), " vmcnt(0):", sum(1 for l in body if "vmcnt(0)" in l))
