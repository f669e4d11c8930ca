#!/usr/bin/env python3
"""Instruction mix of a kernel's basic blocks from hipcc -S output (gfx950).

usage: isa_mix.py file.s kernel_substring [--blocks]
Prints, per basic block (label to label) with >= MIN instructions, the count of MFMA / VALU / transcendental /
LDS / VMEM / SALU / waitcnt instructions.  Used to budget VALU-per-MFMA in the attention loops.
"""
import re, sys, collections

def classify(op):
    if op.startswith('v_mfma') or op.startswith('v_smfma'): return 'mfma'
    if op in ('v_exp_f32', 'v_log_f32', 'v_rcp_f32', 'v_rsq_f32', 'v_sqrt_f32', 'v_sin_f32', 'v_cos_f32'): return 'trans'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_') or op.startswith('scratch_'): return 'vmem'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'branch'
    if op.startswith('s_'): return 'salu'
    return 'other'

def main():
    path, key = sys.argv[1], sys.argv[2]
    minins = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split('\n')
    inside = False
    blocks = []  # (label, Counter, oplist)
    cur = None
    for ln in lines:
        m = re.match(r'^(\S+):', ln)
        if m:
            lab = m.group(1)
            if not lab.startswith('.L'):
                inside = (key in lab) and not lab.endswith('.kd')
                if inside: print('kernel', lab)
            if inside:
                cur = [lab, collections.Counter(), collections.Counter()]
                blocks.append(cur)
            continue
        if not inside or cur is None: continue
        s = ln.strip()
        if not s or s.startswith(';') or s.startswith('.'): continue
        op = s.split()[0]
        if op == 's_endpgm': cur[1]['salu'] += 1
        c = classify(op)
        cur[1][c] += 1
        cur[2][op] += 1
    for lab, cnt, ops in blocks:
        n = sum(cnt.values())
        if n < minins: continue
        print(f'{lab:12s} n={n:4d} ' + ' '.join(f'{k}={v}' for k, v in sorted(cnt.items())))
        if '--ops' in sys.argv:
            print('    ' + ' '.join(f'{k}:{v}' for k, v in ops.most_common(40)))

main()
