#!/usr/bin/env python3
"""Developer tool: one kernel of a saved ISA file cut at its s_barrier instructions: per interval the instruction mix in program order
(MFMAs, other vector ALU, scalar ALU, LDS reads, LDS-DMA loads, stores, waits, branches). usage: tools/isa_intervals.py file.s kernel_substring"""
import re, sys
s = open(sys.argv[1]).read()
name = [m.group(1) for m in re.finditer(r'^(_Z\w+):', s, re.M) if sys.argv[2] in m.group(1)][0]
i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
keys = ["mfma", "valu", "salu", "ds_rd", "ds_wr", "dma", "vload", "store", "wait", "nop", "br", "scratch"]
cur = dict.fromkeys(keys, 0); lab = "entry"; line0 = 0
print(name); print("  line label      " + " ".join("%6s" % k for k in keys))
def flush(n):
    global cur
    if sum(cur.values()): print("  %5d %-10s " % (line0, lab) + " ".join("%6d" % cur[k] for k in keys))
    cur = dict.fromkeys(keys, 0)
for n, ln in enumerate(s[i:j].splitlines()):
    t = ln.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', t)
    if m: lab = m.group(1); continue
    if not t or t.startswith(';') or t.startswith('.'): continue
    op = t.split()[0]
    if op == 's_barrier': flush(n); line0 = n; continue
    if 'v_mfma' in op: cur["mfma"] += 1
    elif op.startswith('scratch_'): cur["scratch"] += 1
    elif op.startswith('buffer_load') and ' lds' in t: cur["dma"] += 1
    elif op.startswith('buffer_load') or op.startswith('global_load'): cur["vload"] += 1
    elif op.startswith('buffer_store') or op.startswith('global_store'): cur["store"] += 1
    elif op.startswith('ds_read') or op.startswith('ds_load'): cur["ds_rd"] += 1
    elif op.startswith('ds_'): cur["ds_wr"] += 1
    elif op == 's_waitcnt': cur["wait"] += 1
    elif op == 's_nop': cur["nop"] += 1
    elif op.startswith('s_cbranch') or op == 's_branch': cur["br"] += 1
    elif op.startswith('v_'): cur["valu"] += 1
    elif op.startswith('s_'): cur["salu"] += 1
flush(0)
