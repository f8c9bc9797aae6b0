#!/usr/bin/env python3
"""Compact view of a kernel's ISA: runs of MFMA / VALU / LDS instructions collapsed, every wait, barrier, branch label, direct-to-LDS
load, global load, store and scratch access kept.  usage: trace_isa.py file.s kernel-name-substring [first_line last_line]"""
import re
import sys

s = open(sys.argv[1]).read()
funcs = re.split(r'\n(?=_Z[^\n]*:)', s)
f = [x for x in funcs if sys.argv[2] in x.split(':')[0]][0]
lines = f.split('\n')
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, len(lines))
run = {}
def flush():
    global run
    if run:
        print('      ' + ' '.join(f'{k}x{v}' for k, v in run.items()))
    run = {}
for i, l in enumerate(lines[lo:hi], lo):
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        if re.match(r'\.LBB', t):
            flush(); print(f'{i:6d} {t}')
        continue
    op = t.split()[0]
    if op.startswith('v_mfma'): run['mfma'] = run.get('mfma', 0) + 1
    elif op.startswith('ds_read') or op.startswith('ds_write'): run[op[:8]] = run.get(op[:8], 0) + 1
    elif op.startswith('v_') or op.startswith('s_mov') or op.startswith('s_add') or op.startswith('s_lshl') or op.startswith('s_mul') or op.startswith('s_and') or op.startswith('s_cmp') or op.startswith('s_nop') or op.startswith('s_or') or op.startswith('s_cselect') or op.startswith('s_sub') or op.startswith('s_ashr') or op.startswith('s_lshr') or op.startswith('s_xor') or op.startswith('s_bfe') or op.startswith('s_min') or op.startswith('s_max'):
        k = 'valu' if op.startswith('v_') else 'salu'
        run[k] = run.get(k, 0) + 1
    else:
        flush(); print(f'{i:6d} {t[:110]}')
flush()
