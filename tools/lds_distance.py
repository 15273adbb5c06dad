#!/usr/bin/env python
"""For the largest basic block of a kernel: for every ds_read_b128, how many MFMAs are issued between the read and the first
MFMA that consumes its destination registers (the software-pipelining distance the compiler actually produced).
usage: lds_distance.py file.s kernel_substring"""
import re
import sys

src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and ':' in l and key in l.split(':')[0])
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
blocks, cur = [], []
for l in lines[start + 1:end]:
    t = l.strip()
    if re.match(r'^\.LBB\d+_\d+:', t) or t.startswith('s_cbranch') or t.startswith('s_branch'):
        blocks.append(cur)
        cur = []
        continue
    if t and not t.startswith(';') and not t.startswith('.'):
        cur.append(t)
blocks.append(cur)
body = max(blocks, key=lambda b: sum('v_mfma' in x for x in b))


def regs(tok):
    m = re.match(r'([va])\[(\d+):(\d+)\]', tok)
    if m:
        return {(m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r'([va])(\d+)$', tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


pending, dist, nm = [], [], 0
waits = 0
for ins in body:
    op, _, rest = ins.partition(' ')
    toks = [x.strip() for x in rest.split(',')]
    if op == 'ds_read_b128':
        pending.append((regs(toks[0]), nm))
    elif op.startswith('v_mfma'):
        used = regs(toks[1]) | regs(toks[2])
        for p in list(pending):
            if p[0] & used:
                dist.append(nm - p[1])
                pending.remove(p)
        nm += 1
    elif op == 's_waitcnt' and 'lgkmcnt' in rest:
        waits += 1
print('mfma', nm, 'ds_read_b128 with an MFMA consumer', len(dist), 'lgkm waits', waits)
print('MFMAs between read and first use (histogram):', {d: dist.count(d) for d in sorted(set(dist))})
