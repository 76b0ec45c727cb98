#!/usr/bin/env python3
"""per basic block of one kernel in a saved ISA file: counts of MFMAs, scratch (spill) instructions, LDS reads, buffer stores / loads, barriers
usage: tools/isa_blocks.py file.s kernel_name_substring"""
import re, sys
s = open(sys.argv[1]).read()
names = [m.group(1) for m in re.finditer(r'^(_Z\w+):', s, re.M) if sys.argv[2] in m.group(1)]
for name in names:
    i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
    blk = None; stats = []
    for ln in s[i:j].splitlines():
        m = re.match(r'^(\.LBB\d+_\d+):', ln)
        if m:
            blk = [m.group(1), 0, 0, 0, 0, 0, 0]; stats.append(blk); continue
        if blk is None:
            blk = ['entry', 0, 0, 0, 0, 0, 0]; stats.append(blk)
        if 'v_mfma' in ln: blk[1] += 1
        if 'scratch_' in ln: blk[2] += 1
        if 'ds_read' in ln: blk[3] += 1
        if 'buffer_store' in ln: blk[4] += 1
        if 'buffer_load' in ln: blk[5] += 1
        if 's_barrier' in ln: blk[6] += 1
    print(name)
    print("  block mfma scratch ds_read bstore bload barrier")
    for b in stats:
        if b[1] or b[2] or b[4] or b[6]: print("  ", *b)
