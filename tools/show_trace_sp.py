#!/usr/bin/env python
"""Read /tmp/moe_trace.bin (MOE_DBG=64: s_memtime stamps [wg<8][iteration<32][wave<4][slot<16] of one conv3x3_sp launch) and
print per-iteration medians: body / vmcnt wait / barrier / period, per-k-step cycles (build with -DMOE_STEP_STAMPS) and the
prologue.  With -DMOE_STAMP_MIN only the period is meaningful (no stamp overhead inside the iteration)."""
import numpy as np

t = np.fromfile('/tmp/moe_trace.bin', dtype=np.uint64)[:8 * 32 * 4 * 16].reshape(8, 32, 4, 16).astype(np.int64)
for wg in (0, 3, 5):
    for w in (0, 3):
        n = int((t[wg, :, w, 0] != 0).sum())          # iterations this workgroup actually ran (<= 32 recorded)
        a = t[wg, min(2, max(0, n - 3)):max(n - 1, 1), w]
        if len(a) < 2:
            continue
        per = np.diff(a[:, 0])
        line = 'wg {} wave {} iters {:2d} period={}'.format(wg, w, n, int(np.median(per)))
        if a[:, 1:4].any():
            d = a - a[:, :1]
            line += '  body={} vmwait={} barrier={}'.format(int(np.median(d[:, 1])), int(np.median(d[:, 2] - d[:, 1])), int(np.median(d[:, 3] - d[:, 2])))
        print(line)
        if a[:, 4:14].any():
            st = np.concatenate([a[:, :1], a[:, 4:16]], axis=1)
            print('      per-step cycles (steps 0..11):', ' '.join(str(int(v)) for v in np.median(np.diff(st, axis=1), axis=0)))
pro = t[:, 0, :, 15] - t[:, 0, :, 14]
if pro.any():
    print('prologue cycles per wg (wave 0):', ' '.join(str(int(v)) for v in pro[:, 0]))
import os
if os.environ.get('TRACE_FIRST'):
    for wg in (0, 5):
        a = t[wg, :, 0, 0]
        n = int((a != 0).sum())
        print('wg', wg, 'iteration-start deltas:', ' '.join(str(int(v)) for v in np.diff(a[:n])[:14]), ' prologue', int(t[wg, 0, 0, 15] - t[wg, 0, 0, 14]), ' start->iter0', int(a[0] - t[wg, 0, 0, 14]))
span = t[:, 0, 0, 13] - t[:, 0, 0, 14]
if (t[:, 0, 0, 13] != 0).any():
    print('workgroup start -> end cycles (wg 0..7, wave 0):', ' '.join(str(int(v)) for v in span))
