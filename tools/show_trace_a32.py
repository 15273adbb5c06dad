#!/usr/bin/env python
"""Run the fixed Net4x workload with the -DA32_TRACE library (tools/mk_variant.sh trace32 arsb32c.hip -DA32_TRACE) and print, for arsb32c_kernel, the
cycles each wave spends per phase of a patch iteration (s_memtime ticks = shader cycles)."""
import os
import shutil
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib, trace = os.path.join(ROOT, 'moephoto_amd', 'libmoephoto_amd.so'), os.path.join(ROOT, 'moephoto_amd', '_abl', sys.argv[1] if len(sys.argv) > 1 else 'lib_trace32.so')
shutil.copy(lib, '/tmp/lib_orig.so')
try:
    shutil.copy(trace, lib)
    env = dict(os.environ, MOE_ARSB_TRACE='1', PROF_ITER='1', PROF_B=os.environ.get('PROF_B', '12'))
    subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'prof_workload.py')], env=env, check=True, stdout=subprocess.DEVNULL)
finally:
    shutil.copy('/tmp/lib_orig.so', lib)
raw = open('/tmp/arsb_trace.bin', 'rb').read()
v = struct.unpack('<{}Q'.format(len(raw) // 8), raw)
names = ['top'] + ['c1.s%d' % r for r in range(7)] + ['mrow4+fix', 'wait', 'barB'] + ['c2.s%d' % r for r in range(6)] + ['yrow3']
for g in (0, 3):
    for p in (4, 7):
        print('workgroup {} patch {}'.format(g, p))
        for w in range(4):
            s = v[((g * 16 + p) * 4 + w) * 40:((g * 16 + p) * 4 + w) * 40 + 18]
            nxt = v[((g * 16 + p + 1) * 4 + w) * 40]
            if not s[0]:
                continue
            d = [s[i] - s[i - 1] for i in range(1, 18)] + [nxt - s[17] if nxt else 0]
            print('  wave {}: total {:6d} | '.format(w, (nxt or s[17]) - s[0]) + ' '.join('{}={}'.format(names[i + 1] if i + 1 < len(names) else 'loop', d[i]) for i in range(len(d))))
