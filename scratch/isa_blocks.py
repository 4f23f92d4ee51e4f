#!/usr/bin/env python3
"""Per-basic-block instruction census of one kernel in a hipcc -save-temps .s file: isa_blocks.py file.s first_line last_line"""
import re, sys
from collections import Counter
f, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
body = open(f).read().split('\n')[a:b]
blocks = []; cur = ['entry', []]
for l in body:
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append(cur); cur = [l.split(':')[0], []]
    else:
        cur[1].append(l)
blocks.append(cur)
for name, ls in blocks:
    ins = [l.split()[0] for l in ls if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
    if len(ins) > 40:
        c = Counter(ins)
        g = lambda pre: sum(v for k, v in c.items() if k.startswith(pre))
        print(f"{name:12s} n={len(ins):5d} mfma={g('v_mfma'):3d} scratch={g('scratch_'):3d} accvgpr={g('v_accvgpr'):4d} exp={g('v_exp_f32'):3d} "
              f"max3={g('v_max3'):3d} add={c.get('v_add_f32',0)+c.get('v_pk_add_f32',0):3d} cvt={g('v_cvt'):3d} nop={c.get('s_nop',0):3d} wait={c.get('s_waitcnt',0):3d} "
              f"mov={c.get('v_mov_b32',0)+c.get('v_pk_mov_b32',0):3d} ds={g('ds_'):3d} valu={sum(v for k,v in c.items() if k.startswith('v_') and not k.startswith('v_mfma')):4d}")
