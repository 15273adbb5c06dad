#!/usr/bin/env python
"""Cycles per phase of a patch iteration of conv64_q8_kernel (tools/mk_variant.sh traceq8 conv64_q8.hip -DQ8_TRACE; the library's 40th q8 launch carries the stamps)."""
import os, shutil, struct, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib, trace = os.path.join(ROOT, 'moephoto_amd', 'libmoephoto_amd.so'), os.path.join(ROOT, 'moephoto_amd', '_abl', 'lib_traceq8.so')
shutil.copy(lib, '/tmp/lib_orig_q8.so')
try:
    shutil.copy(trace, lib)
    env = dict(os.environ, MOE_X3_IMPL='q8', TM_ONLY='SR a2', TM_PREC='auto')
    subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'time_models.py')], env=env, check=True, stdout=subprocess.DEVNULL)
finally:
    shutil.copy('/tmp/lib_orig_q8.so', lib)
raw = open('/tmp/q8_trace.bin', 'rb').read()
v = struct.unpack('<{}Q'.format(len(raw) // 8), raw)
names = ['wait+B1', 'cvt_lo', 'B2'] + ['sh.s%d' % s for s in range(6)] + ['radd3', 'B3', 'cvt_hi', 'B4', 'reads'] + ['lg.s%d' % s for s in range(6)] + ['row3']
for g in (0, 3):
    for pi, p in enumerate((4, 7)):
        print('workgroup {} patch {}'.format(g, p))
        for w in range(4):
            s = v[((g * 2 + pi) * 4 + w) * 32:((g * 2 + pi) * 4 + w) * 32 + 22]
            if not s[0]:
                continue
            d = [s[i] - s[i - 1] for i in range(1, 22)]
            print('  wave {}: total {:6d} | '.format(w, s[21] - s[0]) + ' '.join('{}={}'.format(names[i], d[i]) for i in range(len(d))))
