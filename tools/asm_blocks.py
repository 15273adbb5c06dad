#!/usr/bin/env python
"""Instruction histogram of the largest basic blocks of one kernel in a -save-temps .s file.
usage: asm_blocks.py file.s kernel_substring [nblocks]"""
import collections
import re
import sys

src, key = sys.argv[1], sys.argv[2]
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 4
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and ':' in l and key in l.split(':')[0])
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('.end_amdhsa_kernel') or lines[i].strip().startswith('s_endpgm'))
blocks, cur, name = [], [], 'entry'
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        if re.match(r'^\.LBB\d+_\d+:', t):
            blocks.append((name, cur))
            name, cur = t.split(':')[0], []
        continue
    op = t.split()[0]
    cur.append(op)
    if op.startswith('s_cbranch') or op == 's_branch':
        blocks.append((name, cur))
        name, cur = name + '+', []
blocks.append((name, cur))
blocks.sort(key=lambda b: -len(b[1]))
for name, ops in blocks[:nb]:
    h = collections.Counter(ops)
    mf = sum(v for k, v in h.items() if 'mfma' in k)
    print('--- block {} : {} instructions, {} mfma, {:.2f} fillers per mfma'.format(name, len(ops), mf, (len(ops) - mf) / max(1, mf)))
    print('   ' + '  '.join('{}:{}'.format(k, v) for k, v in h.most_common(40)))
