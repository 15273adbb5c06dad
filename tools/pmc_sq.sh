#!/bin/bash
# counters of conv64_q8 / conv64_sq on tools/time_sq.py's launch sets: MFMA busy, VALU share, waits, effective clock
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${PMC_TAG:-r04r}
mkdir -p $OUT
RE='conv64_q8|conv64_sq|arsb_sq'
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY --kernel-include-regex "$RE" -d $OUT/pmc_sq -o pmc -f csv -- python tools/time_sq.py > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SALU --kernel-include-regex "$RE" -d $OUT/pmc_g -o pmc -f csv -- python tools/time_sq.py > $OUT/pmc_g.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/st -o st -f csv -- python tools/time_sq.py > $OUT/st.log 2>&1
python - $OUT <<'P'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for d in ('pmc_sq', 'pmc_g'):
    for f in glob.glob(out + '/' + d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void (anonymous namespace)::', '')
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            if r['Counter_Name'] in ('SQ_WAVE_CYCLES', 'GRBM_GUI_ACTIVE'):
                cnt[(k, r['Counter_Name'])] += 1
dur = {}
for f in glob.glob(out + '/st/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r['Name'].split('(')[0].replace('void (anonymous namespace)::', '')] = float(r['AverageNs'])
for k, c in sorted(agg.items()):
    n = max(1, cnt[(k, 'SQ_WAVE_CYCLES')])
    wc = c['SQ_WAVE_CYCLES']
    gui = c['GRBM_GUI_ACTIVE'] / max(1, cnt[(k, 'GRBM_GUI_ACTIVE')])
    us = dur.get(k, 0) / 1e3
    print('%-34s launches %3d  %.1f us  MFMA busy %.3f  VALU active %.3f  wait-inst %.3f  LDS active %.3f  VMEM active %.3f  SALU %.3f | per wave-step?: VALU insts/MFMA inst %.2f  clock %.2f GHz' % (
        k, n, us, c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * wc), c['SQ_ACTIVE_INST_VALU'] / wc, c['SQ_WAIT_INST_ANY'] / wc, c['SQ_ACTIVE_INST_LDS'] / wc, c['SQ_ACTIVE_INST_VMEM'] / wc,
        c['SQ_ACTIVE_INST_SCA'] / wc, c['SQ_INSTS_VALU'] / max(1, c['SQ_INSTS_MFMA']), gui / max(1e-9, us * 1e3)))
P
