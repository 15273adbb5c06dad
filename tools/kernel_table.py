#!/usr/bin/env python
"""Condense the per-case kernel_stats of tools/kernel_table.sh into the kernel-resolution table (stdout, JSON):
   cases     {"<key>/<precision>/<class>": [library kernel instantiations the case launched, sorted]}
   launched  their union;  compiled: the library's instantiations (nm -C);  never: compiled - launched"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_census as kc  # noqa: E402

comp = kc.compiled()
cases, union = {}, set()
for f in sorted(glob.glob(os.path.join(sys.argv[1], '*__*__*.csv'))):
    key, prec, cls = os.path.basename(f)[:-4].split('__')
    ks = sorted({kc.norm(r['Name']) for r in csv.DictReader(open(f))} & comp)
    cases['{}/{}/{}'.format(key, prec, cls)] = ks
    union |= set(ks)
json.dump({'what': 'kernel instantiations of libmoephoto_amd.so launched per (zoo key / precision / shape class): tools/kernel_table.sh on an MI355X; frame = batched doCrop of 3x300x420 with 256-px tiles, '
                   'tile = the per-tile call (3 planes of 64x96), odd = rows not a multiple of four and tiny shapes (the fast kernels\' fallbacks)',
           'cases': cases, 'launched': sorted(union), 'compiled': sorted(comp), 'never': sorted(comp - union)}, sys.stdout, indent=1)
