#!/usr/bin/env python
"""Condense rocprofv3 csv output (kernel stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, 'stats', '**', '*kernel_stats.csv'), recursive=True)):
    print('== kernel stats:', f)
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print('  {:>9s} calls  total {:>12s} ns  avg {:>12s} ns  {:>6s} %  {}'.format(r.get('Calls', ''), r.get('TotalDurationNs', ''), r.get('AverageNs', ''),
              r.get('Percentage', ''), r.get('Name', '')[:110]))
for d in sorted(glob.glob(os.path.join(out, 'pmc_*'))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    print('== pmc pass', os.path.basename(d), '({} csv)'.format(len(files)))
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r.get('Kernel_Name', '')[:60] + ' grid=' + r.get('Grid_Size', '') + ' lds=' + r.get('LDS_Block_Size', '') + ' vgpr=' + r.get('VGPR_Count', '') + ' agpr=' + r.get('Accum_VGPR_Count', '')
            agg[k][r.get('Counter_Name', '')] += float(r.get('Counter_Value', 0) or 0)
            cnt[(k, r.get('Counter_Name', ''))] += 1
    for k, v in agg.items():
        print('  ', k)
        for c, val in sorted(v.items()):
            n = cnt[(k, c)]
            print('      {:34s} sum {:18.0f}  per-dispatch {:16.1f}  ({} dispatches)'.format(c, val, val / max(1, n), n))
    log = d + '.log'
    if os.path.exists(log):
        tail = open(log).read().strip().splitlines()[-3:]
        print('   log:', ' | '.join(tail)[:300])
