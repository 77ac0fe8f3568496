#!/usr/bin/env python3
"""ISA evidence for csrc/common.hpp store_b128_guarded: in a hipcc -S listing, every 16-byte buffer store is followed
forward until one of its data registers is written; reported per store: the form of its soffset (SGPR / literal `off`), how many
instructions and how many wait states (s_nop N = N+1, every other instruction = 1) lie between the store and the first VALU /
MFMA / VMEM-load write of a data register.  The gfx950 hazard ("VMEM store of more than 64 bits followed by a write of its
vdata VGPRs") needs the writer to be >= 2 wait states behind the store.
usage: store_hazard_isa.py listing.s [listing2.s ...]"""
import collections
import re
import sys

REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def regs(tok):
    m = REG.fullmatch(tok.strip().rstrip(","))
    if not m:
        return set()
    if m.group(1):
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return {int(m.group(3))}


def dest_regs(line):
    """VGPRs an instruction writes (first operand of VALU / MFMA / loads; stores, branches, waits write none)."""
    ins = line.split()
    if not ins:
        return set()
    op = ins[0]
    if op.startswith(("buffer_store", "global_store", "ds_write", "s_", "buffer_atomic", "global_atomic", ";", ".")) or op.endswith(":"):
        return set()
    if not op.startswith(("v_", "buffer_load", "global_load", "ds_read", "ds_bpermute")):
        return set()
    first = line.split(None, 1)[1].split(",")[0] if len(ins) > 1 else ""
    return regs(first)


for path in sys.argv[1:]:
    lines = [ln.strip() for ln in open(path) if ln.strip() and not ln.strip().startswith((";", ".", "//"))]
    stats = collections.Counter()
    worst = {}
    kernel = "?"
    for i, ln in enumerate(lines):
        if ln.endswith(":") and ln.startswith("_Z"):
            kernel = ln[:-1]
        if not ln.startswith("buffer_store_dwordx4"):
            continue
        ops = [t.strip() for t in ln.split(None, 1)[1].split(",")]
        data = regs(ops[0])
        soff = ops[3].split()[0] if len(ops) > 3 else "?"
        form = "soffset=SGPR" if soff.startswith("s") else "soffset=" + soff
        states = 0
        hit = None
        for j in range(i + 1, min(i + 40, len(lines))):
            nxt = lines[j]
            if nxt.endswith(":") or nxt.startswith(("s_branch", "s_cbranch", "s_endpgm")):
                break
            if dest_regs(nxt) & data:
                hit = (j - i - 1, states, nxt.split()[0])
                break
            m = re.match(r"s_nop (\d+)", nxt)
            states += int(m.group(1)) + 1 if m else 1
        if hit is None:
            stats[(form, "data registers not rewritten within the block")] += 1
        else:
            key = (form, "rewritten after %s wait state(s)" % ("0" if hit[1] == 0 else "1" if hit[1] == 1 else ">=2"))
            stats[key] += 1
            if hit[1] < 2 and key not in worst:
                worst[key] = (kernel, i, lines[i:i + hit[0] + 2])
    print(path)
    for k, v in sorted(stats.items()):
        print("   %-14s %-48s %5d stores" % (k[0], k[1], v))
    for k, (kern, i, ctx) in worst.items():
        print("   example (%s, %s) in %s:" % (k[0], k[1], kern[:70]))
        for c in ctx:
            print("        " + c)
