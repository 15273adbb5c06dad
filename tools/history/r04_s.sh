#!/bin/bash
# per-launch durations of the split-operand layers over one a2 1080p frame, q8_impl = s and p (kernel trace)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${R04_TAG:-r04s}
mkdir -p $OUT
for impl in s p; do
  MOE_Q8_IMPL=$impl TM_ONLY="SR a2" TM_PREC=auto timeout 300 rocprofv3 --kernel-trace -d $OUT/tr_$impl -o tr -f csv -- python tools/time_models.py > $OUT/tm_$impl.txt 2>&1
done
python - $OUT <<'P'
import csv, glob, sys, collections, re
out = sys.argv[1]
for impl in 's', 'p':
    f = glob.glob(out + '/tr_%s/**/*kernel_trace.csv' % impl, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if 'conv64_' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # last frame: the last 9 * 5 launches
    per = collections.OrderedDict()
    last = rows[-45:]
    tot = 0
    line = []
    for r in last:
        us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        tot += us
        line.append('%s:%d=%.0f' % (re.search(r'<([^>]*)>', r['Kernel_Name']).group(1).replace(' ', '').replace('true', 't').replace('false', 'f'), int(r['Grid_Size_X']) if 'Grid_Size_X' in r else 0, us))
    print(impl, 'last frame: %d launches, %.1f us total' % (len(last), tot))
    print('  ' + '  '.join(line))
P
