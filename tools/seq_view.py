#!/usr/bin/env python
"""Compact view of the compiled schedule of a kernel's largest basic block:
r = ds_read_b128, M = MFMA, | = s_waitcnt lgkmcnt, D = LDS-DMA, S = global store, L = global load; plus the read->use histogram.
usage: seq_view.py file.s kernel_substring"""
import re
import sys

src = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith('_Z') and ':' in l and key in l.split(':')[0])
end = next(i for i in range(start, len(src)) if src[i].strip().startswith('s_endpgm'))
blocks, cur = [], []
for l in src[start + 1:end]:
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
    return {(m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1)} if m else set()


seq, pending, dist, nm = '', [], [], 0
for ins in body:
    op, _, rest = ins.partition(' ')
    toks = [x.strip() for x in rest.split(',')]
    if op == 'ds_read_b128':
        seq += 'r'
        pending.append((regs(toks[0]), nm))
    elif op.startswith('v_mfma'):
        seq += 'M'
        used = regs(toks[1]) | regs(toks[2])
        for p in list(pending):
            if p[0] & used:
                dist.append(nm - p[1])
                pending.remove(p)
        nm += 1
    elif op == 's_waitcnt' and 'lgkmcnt' in rest:
        seq += '|'
    elif op.startswith('global_load_lds'):
        seq += 'D'
    elif op.startswith('global_store'):
        seq += 'S'
    elif op.startswith('global_load'):
        seq += 'L'
print(seq)
short = sum(1 for d in dist if d < 6)
print('reads', len(dist), 'with <6 MFMAs to first use:', short, ' lgkm waits:', seq.count('|'), ' instructions:', len(body))
