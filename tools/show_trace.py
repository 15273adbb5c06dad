import numpy as np, sys
t = np.fromfile('/tmp/moe_trace.bin', dtype=np.uint64).reshape(8, 32, 2, 8).astype(np.int64)
names = ['start', 'c_end', 'e_math', 'stg', 'dma', 'st', 'vm', 'bar']
for wg in (0, 3, 5):
    acc = {0: [], 1: []}
    for p in range(2, 30):
        for grp in (0, 1):
            s = t[wg, p, grp]
            role = 0 if (p & 1) == grp else 1
            acc[role].append([(s[k] - s[0]) if s[k] else -1 for k in range(8)])
    for role, nm in ((0, 'compute'), (1, 'epilogue')):
        a = np.array(acc[role])
        print('wg', wg, nm, ' '.join('{}={}'.format(names[k], int(np.median(a[:, k]))) for k in range(1, 8) if np.median(a[:, k]) >= 0))
    print('wg', wg, 'phase period', int(np.median(np.diff(t[wg, 2:30, 0, 0]))))
