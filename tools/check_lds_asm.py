#!/usr/bin/env python3
"""Lint the gfx950 ISA of the attention kernels for the one hazard the compiler cannot see.

The column fragments of the attention kernels come from `ds_read_b64_tr_b16` issued by inline asm (the compiler would
otherwise drain the next tile's LDS-DMA in front of every such read); the kernels settle them with counted
`s_waitcnt lgkmcnt(n)` (col_settle in attention.hip).  To the compiler the asm's destination registers are defined the
moment the asm statement ends, so it is free to READ them -- a register copy at a control-flow merge, a spill -- before
the wait.  Round 3 hit exactly that (v_mov_b64 of two fragments in front of their s_waitcnt in the last-tile path:
sporadic garbage output rows at B = 64).  This script walks every kernel's instruction stream in program order, keeps
the queue of outstanding LDS operations (LDS returns in order), retires them at each s_waitcnt lgkmcnt(n), and reports
any instruction that reads or overwrites a destination register of a still-outstanding transposed read.

    hipcc ... -S --cuda-device-only attention.hip -o attention.s ; python tools/check_lds_asm.py attention.s

Control flow is not followed (the scan is linear over the text): a branch target is entered with the queue as the
textually preceding code left it, which is exact for the straight-line tile bodies this guards.  Exit code 1 on a finding.
"""
import re
import sys

REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def check(path):
    findings = 0
    kernel = None
    queue = []          # outstanding LDS ops in issue order: (is_tr, dest_regs, line_no)
    for ln, line in enumerate(open(path), 1):
        s = line.split(";")[0].strip()
        if not s:
            continue
        if s.endswith(":"):
            if not s.startswith(".L"):
                kernel, queue = s[:-1], []
            continue
        if s.startswith("."):
            continue
        parts = s.split(None, 1)
        op, args = parts[0], (parts[1] if len(parts) > 1 else "")
        if op == "s_endpgm":
            queue = []
            continue
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", args)
            if m:
                n = int(m.group(1))
                while len(queue) > n:
                    queue.pop(0)
            continue
        if op == "s_barrier":
            continue
        operands = [a.strip() for a in args.split(",")]
        pending = set()
        for is_tr, d, _ in queue:
            if is_tr:
                pending |= d
        if pending:
            touched = set()
            for a in operands:
                touched |= regs(a)
            hit = touched & pending
            # an LDS read may legitimately name a pending register only as its own destination AFTER the old value retired;
            # anything else touching an in-flight destination is the hazard
            if hit:
                print(f"{path}:{ln}: [{kernel}] `{s}` touches v{sorted(hit)} while an LDS read into them is outstanding")
                findings += 1
        if op.startswith("ds_"):
            dest = regs(operands[0]) if op.startswith("ds_read") or "rtn" in op else set()
            # every LDS read is tracked (round 5: the single-pass backward also issues its ds_read_b128 row fragments from asm; a
            # compiler-issued read never trips this -- the compiler waits before it touches its own destinations)
            queue.append((op.startswith("ds_read"), dest, ln))
        if op.startswith("s_load") or op.startswith("s_buffer_load") or op == "s_memtime":
            queue.append((False, set(), ln))       # SMEM shares lgkmcnt (returns out of order: only ever makes a wait stricter)
    return findings


if __name__ == "__main__":
    bad = sum(check(p) for p in sys.argv[1:])
    print(f"check_lds_asm: {bad} finding(s)")
    sys.exit(1 if bad else 0)
