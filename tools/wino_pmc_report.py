#!/usr/bin/env python
"""Condense the rocprofv3 --pmc passes over tools/micro/wino_probe (tools/r06_*.sh) into a few lines: MFMA busy, what a wave's cycles go to per MFMA, instruction mix,
LDS bank conflicts, effective clock.   python tools/wino_pmc_report.py gpurun_out/<dir>"""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]
K = collections.defaultdict(float)
D = {}
for f in glob.glob(os.path.join(out, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
    name = f.split('pmc_')[1].split('/')[0]
    for r in csv.DictReader(open(f)):
        if 'ILi0E' not in r['Kernel_Name'] and '<0>' not in r['Kernel_Name']:
            continue
        K[r['Counter_Name']] += float(r['Counter_Value'] or 0)
        D.setdefault(name, {})[r['Dispatch_Id']] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
n = {k: len(v) for k, v in D.items()}
print('dispatches per pass', n)
for k in sorted(K):
    print('%-28s %.4g' % (k, K[k]))
if K.get('SQ_WAVE_CYCLES'):
    print('MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_WAVE_CYCLES) = %.3f' % (K['SQ_VALU_MFMA_BUSY_CYCLES'] / 4 / K['SQ_WAVE_CYCLES']))
    print('per MFMA, in cycles (4 x the counters'"'"' quad-cycles; 32 = a busy matrix pipe): wave %.1f = waiting (s_waitcnt / barrier) %.1f + issue stalls %.1f + issuing %.1f' %
          tuple(4 * K[c] / K['SQ_INSTS_MFMA'] for c in ('SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY')))
    print('LDS bank conflict cycles per dispatch %.3g' % (K['SQ_LDS_BANK_CONFLICT'] / n['sq']))
if K.get('SQ_INSTS_VALU') and n.get('sq'):
    m = K['SQ_INSTS_MFMA'] / n['sq'] * n['sq2']
    print('per MFMA: VALU besides the MFMA %.2f, SALU %.2f, LDS %.2f, VMEM rd %.3f; LDS issue stalls %.1f cycles, LDS array active %.1f' %
          ((K['SQ_INSTS_VALU'] / m - 1,) + tuple(K[c] / m for c in ('SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD')) + (4 * K['SQ_WAIT_INST_LDS'] / m, K['SQ_LDS_IDX_ACTIVE'] / m)))
if K.get('GRBM_GUI_ACTIVE'):
    dur = sum(D['grbm'].values())
    print('effective clock = GRBM_GUI_ACTIVE / 8 / duration = %.3f GHz; %.3f ms per dispatch (profiled)' % (K['GRBM_GUI_ACTIVE'] / 8 / dur, dur / n['grbm'] / 1e6))
if K.get('FETCH_SIZE'):
    print('fetch %.3f GB per dispatch (x2: gfx950), write %.3f GB' % (K['FETCH_SIZE'] * 1024 * 2 / n['fetch'] / 1e9, K.get('WRITE_SIZE', 0) * 1024 / max(1, n.get('write', 1)) / 1e9))
