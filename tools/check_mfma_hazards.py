#!/usr/bin/env python3
"""Static check of the compiled gfx950 ISA for matrix-core register hazards the compiler does not see.

The MFMA inner loops are inline asm (only those are checked by (1) and (2)) (in-place accumulators, rsu_common.h), so hipcc's hazard recogniser inserts none of the
wait states the ISA requires around them. This script walks the control-flow graph of every kernel in the given .s files
and reports
  (1) any non-MFMA instruction that touches the destination registers of an MFMA fewer than RESULT_WS wait states after it
      (XDL write -> VALU/VMEM/LDS read or write; also covers the SrcC write-after-read window because SrcC == vDst), and
  (2) any VALU instruction that writes a register an MFMA reads fewer than OPERAND_WS wait states later, and
  (3) any VALU instruction that overwrites the data registers of a 96/128-bit VMEM store fewer than STORE_WS wait states after it.
      hipcc only separates the two when the store has no SGPR offset (the documented hazard); on gfx950 the store's last
      lanes were observed to pick up the new register value with an SGPR offset as well (the epilogue stores of
      igemm_fwd2, once per ~1e4 launches), so every wide store is checked here whatever its addressing.
  (4) any inline-asm VMEM instruction that reads an SGPR a VALU instruction (v_readlane, v_readfirstlane, v_cmp ...) wrote fewer
      than SGPR_WS wait states earlier -- e.g. a spilled descriptor word restored right in front of an asm store.
  (5) any instruction that touches the destination register of an inline-asm ds_bpermute_b32 before an s_waitcnt lgkmcnt(n) has retired it
      (LDS operations return in order: the exchange is back once at most as many LDS / scalar-memory operations as were issued behind it
      may still be outstanding). hipcc does not count LDS operations issued from asm, so a copy it schedules between the exchange and our
      counted wait would read a register whose LDS return is still pending (the transposed epilogue of igemm_pp).
A wait state is one issued instruction of the same wave (s_nop N counts N+1); instructions of other waves do not count, so the
check is conservative. usage: check_mfma_hazards.py file.s [...]; exit status 1 when anything is reported."""
import re
import sys

RESULT_WS = 16   # required by the ISA for the 16x16x32 bf16 MFMA: <= 12
RESULT_WS_32 = 20  # ... for the 32x32x16 form (8 passes of 4 cycles): <= 19
OPERAND_WS = 3   # VALU write -> MFMA read: 2
STORE_WS = 2     # wide VMEM store -> VALU overwrite of its data registers (see (3) below); documented: 1
SGPR_WS = 5      # VALU write of an SGPR -> (inline-asm) VMEM instruction that reads it; documented: 5

REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")
SREG = re.compile(r"\bs(?:(\d+)|\[(\d+):(\d+)\])")
LABEL = re.compile(r"^([.\w$]+):")


def regs_of(operand):
    out = set()
    for m in REG.finditer(operand):
        lo = int(m.group(2) if m.group(2) is not None else m.group(3))
        hi = int(m.group(2) if m.group(2) is not None else m.group(4))
        out.update((m.group(1), r) for r in range(lo, hi + 1))
    return out


def sregs_of(operand):
    out = set()
    for m in SREG.finditer(operand):
        lo = int(m.group(1) if m.group(1) is not None else m.group(2))
        hi = int(m.group(1) if m.group(1) is not None else m.group(3))
        out.update(range(lo, hi + 1))
    return out


class Ins:
    __slots__ = ("op", "ops", "all", "line", "text", "asm")

    def __init__(self, text, line, in_asm=False):
        self.text, self.line, self.asm = text, line, in_asm
        parts = text.split(None, 1)
        self.op = parts[0]
        self.ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        self.all = set()
        for o in self.ops:
            self.all |= regs_of(o)

    def ws(self):
        return int(self.ops[0], 0) + 1 if self.op == "s_nop" else 1

    def is_mfma(self):
        return self.op.startswith("v_mfma") or self.op.startswith("v_smfma")

    def wide_store_data(self):
        if self.op.startswith(("buffer_store_dwordx3", "buffer_store_dwordx4")):
            return regs_of(self.ops[0])
        if self.op.startswith(("global_store_dwordx3", "global_store_dwordx4", "flat_store_dwordx3", "flat_store_dwordx4",
                               "scratch_store_dwordx3", "scratch_store_dwordx4")):
            return regs_of(self.ops[1])
        return set()

    def is_vmem(self):
        return self.op.startswith(("buffer_", "global_", "flat_", "scratch_"))

    def valu_sgpr_writes(self):
        if not self.op.startswith("v_") or not self.ops:
            return set()
        w = sregs_of(self.ops[0])
        if self.op.endswith("_e64") and len(self.ops) > 1 and self.op.startswith(("v_add_co", "v_sub_co", "v_addc", "v_subb", "v_div_scale", "v_mad_u64", "v_mad_i64")):
            w |= sregs_of(self.ops[1])
        return w

    def valu_writes(self):
        if not self.op.startswith("v_") or self.is_mfma() or not self.ops:
            return set()
        if self.op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
            return set()
        return regs_of(self.ops[0])


def kernels(path):
    """yield (name, [Ins], {label: index}) per function"""
    name, ins, labels, in_asm = None, [], {}, False
    for ln, raw in enumerate(open(path), 1):
        if "#ASMSTART" in raw:
            in_asm = True
        elif "#ASMEND" in raw:
            in_asm = False
        line = raw.split(";")[0].rstrip()
        m = LABEL.match(line)
        if m:
            lab = m.group(1)
            if not lab.startswith(".L"):
                if name and ins:
                    yield name, ins, labels
                name, ins, labels = lab, [], {}
            else:
                labels[lab] = len(ins)
            continue
        t = line.strip()
        if not t or t.startswith(".") or not raw.startswith(("\t", " ")):
            continue
        if name:
            ins.append(Ins(t, ln, in_asm))
            if t.startswith("s_endpgm"):
                yield name, ins, labels
                name, ins, labels = None, [], {}


def successors(ins, labels, i):
    op = ins[i].op
    if op == "s_endpgm":
        return []
    if op == "s_branch":
        return [labels[ins[i].ops[0]]] if ins[i].ops[0] in labels else []
    out = [i + 1] if i + 1 < len(ins) else []
    if op.startswith("s_cbranch") and ins[i].ops and ins[i].ops[-1] in labels:
        out.append(labels[ins[i].ops[-1]])
    return out


def walk(ins, labels, start, budget, visit):
    """visit(j, elapsed) for every instruction reachable from start within budget wait states"""
    best = {}
    stack = [(s, 0) for s in successors(ins, labels, start)]
    while stack:
        j, el = stack.pop()
        if el >= budget or best.get(j, 1 << 30) <= el:
            continue
        best[j] = el
        if visit(j, el) is False:
            continue
        for s in successors(ins, labels, j):
            stack.append((s, el + ins[j].ws()))


LGKM = re.compile(r"lgkmcnt\((\d+)\)")
BPERM_BUDGET = 600   # instructions followed behind an exchange before giving up (an epilogue holds ~300)


def check_bpermute(path, name, ins, labels, i):
    """rule (5) for the asm ds_bpermute_b32 at index i; returns the number of violations"""
    dst = regs_of(ins[i].ops[0])
    bad = 0
    seen = {}
    stack = [(s, 0, 0) for s in successors(ins, labels, i)]   # (index, LDS / SMEM operations issued behind the exchange, steps)
    while stack:
        j, later, steps = stack.pop()
        if steps > BPERM_BUDGET or seen.get(j, -1) >= later:
            continue
        seen[j] = later
        J = ins[j]
        if J.op == "s_waitcnt":
            m = LGKM.search(J.text)
            if m is not None and int(m.group(1)) <= later:
                continue   # retired on this path
        elif J.op.startswith(("ds_", "s_load", "s_buffer_load")):
            if J.all & dst:
                print("%s:%d %s: '%s' touches the result of the exchange in line %d before its wait" % (path, J.line, name[:60], J.text, ins[i].line))
                bad += 1
                continue
            later += 1
        elif J.all & dst:
            print("%s:%d %s: '%s' touches the result of the exchange in line %d before its wait" % (path, J.line, name[:60], J.text, ins[i].line))
            bad += 1
            continue
        for s2 in successors(ins, labels, j):
            stack.append((s2, later, steps + 1))
    return bad


def check(path):
    bad = 0
    for name, ins, labels in kernels(path):
        n_mfma = n_store = 0
        for i, I in enumerate(ins):
            if I.asm and I.op == "ds_bpermute_b32":
                bad += check_bpermute(path, name, ins, labels, i)
                continue
            if I.is_mfma() and not I.asm:
                continue  # a builtin MFMA: the compiler's own hazard recogniser covers it
            if I.is_mfma():
                n_mfma += 1
                dst = regs_of(I.ops[0])
                if regs_of(I.ops[3]) != dst and regs_of(I.ops[3]):
                    print("%s:%d %s: accumulator not in place: %s" % (path, I.line, name[:60], I.text))
                    bad += 1

                def visit(j, el, dst=dst, I=I):
                    nonlocal bad
                    J = ins[j]
                    if J.is_mfma():
                        # a later MFMA that redefines the same accumulator takes over the check
                        return not (regs_of(J.ops[0]) == dst)
                    if J.all & dst:
                        print("%s:%d %s: '%s' touches MFMA results %d wait states after line %d" % (
                            path, J.line, name[:60], J.text, el, I.line))
                        bad += 1
                        return False
                    return True
                walk(ins, labels, i, RESULT_WS_32 if "32x32" in I.text else RESULT_WS, visit)
            else:
                sd = I.wide_store_data()
                if sd:
                    def visit3(j, el, sd=sd, I=I):
                        nonlocal bad
                        J = ins[j]
                        if J.valu_writes() & sd:
                            print("%s:%d %s: '%s' overwrites store data %d wait states after line %d" % (
                                path, J.line, name[:60], J.text, el, I.line))
                            bad += 1
                            return False
                        return True
                    walk(ins, labels, i, STORE_WS, visit3)
                    n_store += 1
                sw = I.valu_sgpr_writes()
                if sw:
                    def visit4(j, el, sw=sw, I=I):
                        nonlocal bad
                        J = ins[j]
                        if J.asm and J.is_vmem() and any(sregs_of(o) & sw for o in J.ops):
                            print("%s:%d %s: asm '%s' reads an SGPR written by VALU line %d only %d wait states earlier" % (
                                path, J.line, name[:60], J.text, I.line, el))
                            bad += 1
                            return False
                        return True
                    walk(ins, labels, i, SGPR_WS, visit4)
                w = I.valu_writes()
                if not w:
                    continue

                def visit2(j, el, w=w, I=I):
                    nonlocal bad
                    J = ins[j]
                    if J.is_mfma() and J.asm and (J.all & w):
                        print("%s:%d %s: MFMA reads %s written by VALU line %d only %d wait states earlier" % (
                            path, J.line, name[:60], sorted(J.all & w)[:2], I.line, el))
                        bad += 1
                        return False
                    return not (J.valu_writes() & w)
                walk(ins, labels, i, OPERAND_WS, visit2)
        if n_mfma or n_store:
            print("%-100s %5d MFMA %4d wide stores checked" % (name[:100], n_mfma, n_store))
    return bad


if __name__ == "__main__":
    total = sum(check(p) for p in sys.argv[1:])
    print("hazards reported:", total)
    sys.exit(1 if total else 0)
