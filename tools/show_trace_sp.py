import numpy as np
t = np.fromfile('/tmp/moe_trace.bin', dtype=np.uint64)[:8*32*4*16].reshape(8, 32, 4, 16).astype(np.int64)
for wg in (0, 3, 5):
    for w in (0, 3):
        a = t[wg, 2:30, w]
        d = a - a[:, :1]
        per = np.diff(a[:, 0])
        print('wg', wg, 'wave', w, 'body={} vmwait={} barrier={}  period={}'.format(int(np.median(d[:, 1])), int(np.median(d[:, 2] - d[:, 1])), int(np.median(d[:, 3] - d[:, 2])), int(np.median(per))))
        if a[:, 4:].any():
            st = np.concatenate([a[:, :1], a[:, 4:16]], axis=1)
            print('      per-step cycles (steps 0..11):', ' '.join(str(int(v)) for v in np.median(np.diff(st, axis=1), axis=0)))
pro = t[:, 0, :, 15] - t[:, 0, :, 14]
if pro.any():
    print('prologue cycles per wg (wave 0):', ' '.join(str(int(v)) for v in pro[:, 0]))
