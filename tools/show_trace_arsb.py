#!/usr/bin/env python
"""Run the fixed Net4x workload with the -DARSB_TRACE library (tools/trace_arsb.sh) and print, for the fused ARSB kernel, the
cycles each wave spends per phase of a patch iteration (s_memtime ticks = shader cycles)."""
import os
import shutil
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib, trace = os.path.join(ROOT, 'moephoto_amd', 'libmoephoto_amd.so'), os.path.join(ROOT, 'moephoto_amd', '_abl', sys.argv[1] if len(sys.argv) > 1 else 'lib_trace.so')
shutil.copy(lib, '/tmp/lib_orig.so')
try:
    shutil.copy(trace, lib)
    env = dict(os.environ, MOE_ARSB_TRACE='1', PROF_ITER='1', PROF_B=os.environ.get('PROF_B', '12'))
    subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'prof_workload.py')], env=env, check=True, stdout=subprocess.DEVNULL)
finally:
    shutil.copy('/tmp/lib_orig.so', lib)
raw = open('/tmp/arsb_trace.bin', 'rb').read()
v = struct.unpack('<{}Q'.format(len(raw) // 8), raw)
names = ['top', 'vmcnt0', 'barrier1'] + ['c1.r%d' % r for r in range(12)] + ['m9+border', 'lgkm0', 'barrier2'] + ['c2.r%d' % r for r in range(10)] + ['out7']
for g in (0,):
    for p in (5, 8):
        print('workgroup {} patch {}'.format(g, p))
        for w in range(4):
            s = v[((g * 16 + p) * 4 + w) * 40:((g * 16 + p) * 4 + w) * 40 + 29]
            nxt = v[((g * 16 + p + 1) * 4 + w) * 40]
            if not s[0]:
                continue
            d = [s[i] - s[i - 1] for i in range(1, 29)] + [nxt - s[28] if nxt else 0]
            print('  wave {}: total {:6d} | '.format(w, (nxt or s[28]) - s[0]) + ' '.join('{}={}'.format(names[i + 1] if i + 1 < len(names) else 'loop', d[i]) for i in range(len(d))))
