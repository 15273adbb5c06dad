#!/usr/bin/env python
"""Condense the rocprofv3 PMC passes of tools/pmc_bench.sh (bench.py's own launches) into one JSON + a text table.

    python tools/pmc_collect.py gpurun_out/<tag> <frames profiled>   -> <tag>/pmc_bench.json (copy to profiles/pmc_bench.json)

Per kernel name: dispatches per frame, FETCH_SIZE (doubled: gfx950 tallies 128-byte requests at 64 B for wide coalesced reads,
MI355X_MICROARCH.md section HBM) and WRITE_SIZE in bytes per frame, MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_WAVE_CYCLES)
(wave cycles are counted in quad-cycles, one wave per SIMD in these kernels), LDS bank-conflict cycles, and the effective clock =
GRBM_GUI_ACTIVE / 8 / kernel duration (the counter is summed over the 8 XCDs).  `groups` maps bench.py's bracketed layer keys onto the kernels that run them.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out, frames = sys.argv[1], max(1, int(sys.argv[2]) if len(sys.argv) > 2 else 3)
K = defaultdict(lambda: defaultdict(float))       # kernel -> counter -> sum over dispatches
N = defaultdict(lambda: defaultdict(int))         # kernel -> counter -> dispatches
DUR = defaultdict(lambda: defaultdict(float))     # kernel -> pass -> summed duration (ns)
for d in sorted(glob.glob(os.path.join(out, 'pmc_*'))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)[4:]
    seen = set()
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r'^void ', '', r['Kernel_Name']).replace('(anonymous namespace)::', '')
            k = re.sub(r'\(.*\)$', '', k)
            c = r['Counter_Name']
            K[k][c] += float(r['Counter_Value'] or 0)
            N[k][c] += 1
            key = (k, r['Dispatch_Id'])
            if key not in seen:
                seen.add(key)
                DUR[k][name] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])

kernels = {}
for k in sorted(K, key=lambda k: -DUR[k].get('grbm', DUR[k].get('sq', 0))):
    c, n = K[k], N[k]
    e = {'dispatches_per_frame': round(max(n.values()) / frames, 2)}
    if 'FETCH_SIZE' in c:
        e['fetch_bytes_per_frame'] = int(c['FETCH_SIZE'] * 1024 * 2 / frames)
    if 'WRITE_SIZE' in c:
        e['write_bytes_per_frame'] = int(c['WRITE_SIZE'] * 1024 / frames)
    if 'fetch_bytes_per_frame' in e and 'write_bytes_per_frame' in e:
        e['hbm_bytes_per_frame'] = e['fetch_bytes_per_frame'] + e['write_bytes_per_frame']
    if c.get('SQ_WAVE_CYCLES'):
        e['mfma_busy'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4.0 * c['SQ_WAVE_CYCLES']), 4)
        e['wait_inst_any'] = round(c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES'], 4)
        e['wait_any'] = round(c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'], 4)
        e['lds_bank_conflict_per_dispatch'] = round(c['SQ_LDS_BANK_CONFLICT'] / max(1, n['SQ_LDS_BANK_CONFLICT']), 1)
        e['mfma_insts_per_frame'] = int(c['SQ_INSTS_MFMA'] / frames)
    if c.get('GRBM_GUI_ACTIVE') and DUR[k].get('grbm'):
        e['clock_ghz_profiled'] = round(c['GRBM_GUI_ACTIVE'] / 8.0 / DUR[k]['grbm'], 3)
        e['ms_per_frame_profiled'] = round(DUR[k]['grbm'] / frames / 1e6, 3)
    kernels[k] = e

GROUP_KERNELS = {      # bench.py layer key -> regex of the kernel(s) that run it in the default build
    'convt_R1.up1': r'conv3x3_(sp_kernel<7>|rw_kernel<7[,>]|ps4_kernel<2,)',
    'u.up1': r'conv3x3_(sp_kernel<3>|rw_kernel<3[,>]|ps4_kernel<1,)',
    'arsb': r'arsb(32c?|_fused)_kernel',
    'exact': r'(conv64_(sq|q8|x3)|arsb_sq)_kernel',
}
groups = {}
for key, rx in GROUP_KERNELS.items():
    ks = [k for k in kernels if re.search(rx, k) and 'hbm_bytes_per_frame' in kernels[k]]
    if not ks:
        continue
    g = {'kernels': ks, 'hbm_bytes_per_frame': sum(kernels[k]['hbm_bytes_per_frame'] for k in ks),
         'launches_per_frame': sum(kernels[k]['dispatches_per_frame'] for k in ks),
         'note': 'rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes over bench.py --steps 2 (tools/pmc_bench.sh), all launches of the frame'}
    mb = [kernels[k]['mfma_busy'] for k in ks if 'mfma_busy' in kernels[k]]
    if mb:
        g['mfma_busy'] = round(sum(mb) / len(mb), 4)
    ck = [kernels[k]['clock_ghz_profiled'] for k in ks if 'clock_ghz_profiled' in kernels[k]]
    if ck:
        g['clock_ghz_profiled'] = round(sum(ck) / len(ck), 3)
    groups[key] = g
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from moephoto_amd.build import source_digest  # noqa: E402
json.dump({'frames_profiled': frames, 'source_sha256': source_digest(), 'kernels': kernels, 'groups': groups}, open(os.path.join(out, 'pmc_bench.json'), 'w'), indent=1)
print('%-44s %6s %10s %10s %8s %8s %8s %8s' % ('kernel', 'n/frm', 'fetch MB', 'write MB', 'MFMA', 'waitI', 'GHz', 'ms/frm'))
for k, e in kernels.items():
    print('%-44s %6.1f %10.1f %10.1f %8s %8s %8s %8s' % (k[:44], e['dispatches_per_frame'], e.get('fetch_bytes_per_frame', 0) / 1e6, e.get('write_bytes_per_frame', 0) / 1e6,
          e.get('mfma_busy', ''), e.get('wait_inst_any', ''), e.get('clock_ghz_profiled', ''), e.get('ms_per_frame_profiled', '')))
