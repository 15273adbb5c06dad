#!/usr/bin/env python
"""Run the fixed Net4x workload with the -DPC_TRACE library (tools/trace_pc.sh) and print, for arsb_pc_kernel, the cycles each wave
spends computing a step (barrier release -> arrival at the next barrier) and waiting in the barrier (s_memtime ticks).
Waves 0, 1 = producers (conv_1), 2, 3 = consumers (conv_2, residual, stores, DMA)."""
import os
import shutil
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib, trace = os.path.join(ROOT, 'moephoto_amd', 'libmoephoto_amd.so'), os.path.join(ROOT, 'moephoto_amd', '_abl', sys.argv[1] if len(sys.argv) > 1 else 'lib_pc_trace.so')
shutil.copy(lib, '/tmp/lib_orig.so')
try:
    shutil.copy(trace, lib)
    env = dict(os.environ, MOE_ARSB_TRACE='1', MOE_ARSB_IMPL='pc', PROF_ITER='1', PROF_B=os.environ.get('PROF_B', '12'))
    subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'prof_workload.py')], env=env, check=True, stdout=subprocess.DEVNULL)
finally:
    shutil.copy('/tmp/lib_orig.so', lib)
raw = open('/tmp/arsb_trace.bin', 'rb').read()
v = struct.unpack('<{}Q'.format(len(raw) // 8), raw)
for g in (0,):
    for p in (5, 8):
        print('workgroup {} period {}   (per step: compute / barrier wait)'.format(g, p))
        for w in range(4):
            base = ((g * 16 + p) * 4 + w) * 40
            s, prev = v[base:base + 24], v[((g * 16 + p - 1) * 4 + w) * 40 + 23]
            if not s[0] or not prev:
                continue
            cells, t = [], prev
            for k in range(12):
                cells.append('{}/{}'.format(s[2 * k] - t, s[2 * k + 1] - s[2 * k]))
                t = s[2 * k + 1]
            print('  wave {} ({}): period {:6d} | '.format(w, 'PC'[w >> 1], s[23] - prev) + ' '.join('s{}={}'.format(k, c) for k, c in enumerate(cells)))
            if w >= 2 and v[base + 24]:          # -DPC_SUBSTEP build: cycles per chunk of that step
                ss = int(os.environ.get('PC_SUBSTEP', '8'))
                t0 = s[2 * ss - 1] if ss else prev
                sub = [t0] + [v[base + 24 + i] for i in range(12)] + [s[2 * ss]]
                print('      step {} chunks: '.format(ss) + ' '.join(str(sub[i + 1] - sub[i]) for i in range(12)) + ' | end wait {}'.format(sub[13] - sub[12]))
