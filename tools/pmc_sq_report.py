#!/usr/bin/env python
"""condense tools/pmc_sq.sh's CSVs: per kernel MFMA busy, VALU / LDS / VMEM / SALU issue shares, waits, VALU instructions per MFMA, clock (GRBM_GUI_ACTIVE is summed over the 8 XCDs)"""
import csv, glob, sys, collections, re
out = sys.argv[1]
def key(n):
    m = re.search(r'((?:conv64|arsb)_\w+<[^>]*>)', n)
    return m.group(1) if m else n[:40]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for d in ('pmc_sq', 'pmc_g'):
    for f in glob.glob(out + '/' + d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = key(r['Kernel_Name'])
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            if r['Counter_Name'] in ('SQ_WAVE_CYCLES', 'GRBM_GUI_ACTIVE'):
                cnt[(k, r['Counter_Name'])] += 1
dur = {}
for f in glob.glob(out + '/st/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[key(r['Name'])] = float(r['AverageNs'])
for k, c in sorted(agg.items()):
    n = max(1, cnt[(k, 'SQ_WAVE_CYCLES')])
    wc = c['SQ_WAVE_CYCLES']
    gui = c['GRBM_GUI_ACTIVE'] / max(1, cnt[(k, 'GRBM_GUI_ACTIVE')])
    us = dur.get(k, 0) / 1e3
    print('%-34s n %3d %6.1f us  MFMA busy %.3f  VALU act %.3f  wait-inst %.3f  LDS act %.3f  VMEM act %.3f  SALU %.3f | VALU/MFMA insts %.2f  clock %.2f GHz' % (
        k, n, us, c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * wc), c['SQ_ACTIVE_INST_VALU'] / wc, c['SQ_WAIT_INST_ANY'] / wc, c['SQ_ACTIVE_INST_LDS'] / wc, c['SQ_ACTIVE_INST_VMEM'] / wc,
        c['SQ_ACTIVE_INST_SCA'] / wc, c['SQ_INSTS_VALU'] / max(1, c['SQ_INSTS_MFMA']), gui / 8 / max(1e-9, us * 1e3)))
