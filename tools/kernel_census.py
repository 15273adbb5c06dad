#!/usr/bin/env python
"""Which of the library's compiled kernel instantiations does a workload launch?  ("dead instantiations visible".)

    rocprofv3 --kernel-trace --stats -d OUT -o census -f csv -- <workload>;  python tools/kernel_census.py OUT/census_kernel_stats.csv [label]

Compiled = the host launch stubs of libmoephoto_amd.so (nm -C: one per template instantiation).  Prints the instantiations the workload launched (calls, total ms) and the
ones it never did.  Workloads of tools/history/r05_census.sh: (a) every zoo key in its default arithmetic over a 1080p frame + the per-tile loop + the I/O edges + resize -- the
DEFAULTS; (b) the whole GPU test suite -- everything any test reaches (options, fallbacks, debug paths)."""
import csv
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'moephoto_amd', 'libmoephoto_amd.so')


def norm(name):
    name = re.sub(r'^void ', '', name.strip()).replace('(anonymous namespace)::', '')
    m = re.search(r'([a-z][a-z0-9]*(?:_[a-z0-9]+)*_kernel)I(.+?)E+v', name)      # a name the demangler left alone (template arguments with _Float16): <kernel>I<args>E..v<params>
    if m and '<' not in name:
        args = []
        for tok in re.findall(r'DF16_|L[ib]\d+|[a-z]', m.group(2)):
            args.append('half' if tok == 'DF16_' else ({'f': 'float', 'h': 'unsigned char', 't': 'unsigned short'}.get(tok, tok) if len(tok) == 1 else tok[2:]))
        return '{}<{}>'.format(m.group(1), ', '.join(args))
    m = re.search(r'([a-z][a-z0-9]*(?:_[a-z0-9]+)*_kernel)E', name)
    if m and '<' not in name and '(' not in name:
        return m.group(1)
    name = re.sub(r'\(.*$', '', name)
    return name.replace('_Float16', 'half')


def compiled():
    out = subprocess.run(['nm', '-C', LIB], capture_output=True, text=True).stdout
    ks = set()
    for l in out.splitlines():
        if '__device_stub__' in l:
            ks.add(norm(l.split('__device_stub__', 1)[1]))
    return ks


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    label = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    seen = {}
    for r in rows:
        k = norm(r['Name'])
        c = seen.setdefault(k, [0, 0.0])
        c[0] += int(r['Calls'])
        c[1] += float(r['TotalDurationNs']) / 1e6
    comp = compiled()
    ours = {k: v for k, v in seen.items() if k in comp}
    print('== {}: {} of the library\'s {} kernel instantiations launched'.format(label, len(ours), len(comp)))
    for k in sorted(ours, key=lambda k: -ours[k][1]):
        print('   {:58s} calls {:7d}  total {:10.3f} ms'.format(k, ours[k][0], ours[k][1]))
    dead = sorted(comp - set(ours))
    print('-- never launched by this workload ({}):'.format(len(dead)))
    for k in dead:
        print('   ' + k)
    other = sorted(k for k in seen if k not in comp and not k.startswith(('at::', 'void at::', '__amd', 'ncclDev', 'rccl')) and 'at::native' not in k and 'elementwise' not in k)
    if other:
        print('-- launched, not matched to a library stub (name normalisation): ' + '; '.join(other[:12]))


if __name__ == '__main__':
    main()
