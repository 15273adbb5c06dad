#!/usr/bin/env python
"""Instruction mix per MFMA of the bench's MFMA kernels from the rocprofv3 PMC passes of tools/history/r05_a.sh (pmc_mix / pmc_act / pmc_lds / pmc_grbm): per kernel the
counters summed over its dispatches, normalised by SQ_INSTS_MFMA (pass mix) or by SQ_WAVE_CYCLES where that pass has it -- where a kernel's issue slots (and joules)
go besides the matrix pipe.  Counters a pass failed on (unknown on this rocprofv3) are simply absent.

    python tools/pmc_mix.py gpurun_out/r05a"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]
K = defaultdict(lambda: defaultdict(float))
DUR = defaultdict(lambda: defaultdict(float))
for d in sorted(glob.glob(os.path.join(out, 'pmc_*'))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)[4:]
    seen = set()
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r'^void ', '', r['Kernel_Name']).replace('(anonymous namespace)::', '')
            k = re.sub(r'\(.*\)$', '', k)
            K[k][name + ':' + r['Counter_Name']] += float(r['Counter_Value'] or 0)
            key = (k, r['Dispatch_Id'])
            if key not in seen:
                seen.add(key)
                DUR[k][name] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
for k in sorted(K, key=lambda k: -sum(DUR[k].values())):
    c = K[k]
    print(k)
    mf = c.get('mix:SQ_INSTS_MFMA')
    if mf:
        print('   per MFMA:  VALU (incl. MFMA) {:.2f}  LDS {:.2f}  SALU {:.2f}  VMEM_RD {:.3f}  VMEM_WR {:.3f}   | MFMAs {:.3e}  wave quad-cycles per MFMA {:.2f}'.format(
            c.get('mix:SQ_INSTS_VALU', 0) / mf, c.get('mix:SQ_INSTS_LDS', 0) / mf, c.get('mix:SQ_INSTS_SALU', 0) / mf, c.get('mix:SQ_INSTS_VMEM_RD', 0) / mf,
            c.get('mix:SQ_INSTS_VMEM_WR', 0) / mf, mf, c.get('mix:SQ_WAVE_CYCLES', 0) / mf))
    for p in ('act', 'lds'):
        keys = sorted(n for n in c if n.startswith(p + ':'))
        if keys:
            print('   pass {}: '.format(p) + '  '.join('{} {:.4g}'.format(n.split(':')[1].replace('SQ_', ''), c[n]) for n in keys))
    wc = c.get('lds:SQ_ACTIVE_INST_ANY')
    if c.get('grbm:GRBM_GUI_ACTIVE') and DUR[k].get('grbm'):
        print('   clock {:.3f} GHz over {:.3f} ms (grbm pass)'.format(c['grbm:GRBM_GUI_ACTIVE'] / 8.0 / DUR[k]['grbm'], DUR[k]['grbm'] / 1e6))
