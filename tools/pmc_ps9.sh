#!/bin/bash
# counters of conv3x3_ps9 on an a3 1080p frame (tools/time_models.py): MFMA busy, instruction mix, waits, effective clock -- beside conv3x3_ps4 on an a4 frame, same box
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${PMC_TAG:-r06t}
mkdir -p $OUT
for key in a3 a4; do
  RE='conv3x3_ps9|conv3x3_ps4'
  TM_ONLY="SR $key" TM_PREC=auto timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY --kernel-include-regex "$RE" -d $OUT/pmc_sq_$key -o pmc -f csv -- python tools/time_models.py > $OUT/pmc_sq_$key.log 2>&1
  TM_ONLY="SR $key" TM_PREC=auto timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_INSTS_SALU SQ_WAVES --kernel-include-regex "$RE" -d $OUT/pmc_g_$key -o pmc -f csv -- python tools/time_models.py > $OUT/pmc_g_$key.log 2>&1
  TM_ONLY="SR $key" TM_PREC=auto timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/st_$key -o st -f csv -- python tools/time_models.py > $OUT/st_$key.log 2>&1
done
python - $OUT <<'P'
import csv, glob, sys, collections, re
out = sys.argv[1]
def key(n):
    m = re.search(r'(conv3x3_ps\d_kernel<[^>]*>)', n)
    return m.group(1) if m else None
# only the large launches (the 84-plane launch set): a kernel dispatch with >= 200 workgroups ... keep every dispatch, weight by cycles
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = key(r['Kernel_Name'])
        if not k: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] in ('SQ_WAVE_CYCLES', 'GRBM_GUI_ACTIVE'): cnt[(k, r['Counter_Name'])] += 1
dur = {}
for f in glob.glob(out + '/st_*/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = key(r['Name'])
        if k: dur[k] = (float(r['TotalDurationNs']), int(r['Calls']))
for k, c in sorted(agg.items()):
    wc = c['SQ_WAVE_CYCLES']; mf = max(1.0, c['SQ_INSTS_MFMA'])
    tot_ns, calls = dur.get(k, (0.0, 1))
    gui = c['GRBM_GUI_ACTIVE'] / 8.0                      # summed over the 8 XCDs
    print('%-34s launches %3d  total %8.2f ms | MFMA busy (per wave-cycle / 4) %.3f  wave cycles per MFMA %.1f  VALU/MFMA %.2f  LDS/MFMA %.2f  SALU/MFMA %.2f  VMEM/MFMA %.3f | wait-inst %.3f  wait-any %.3f  wait-LDS %.3f | clock %.2f GHz' % (
        k, calls, tot_ns / 1e6, c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * wc) if wc else 0, wc / mf, c['SQ_INSTS_VALU'] / mf, c['SQ_INSTS_LDS'] / mf, c['SQ_INSTS_SALU'] / mf, c['SQ_INSTS_VMEM'] / mf,
        c['SQ_WAIT_INST_ANY'] / wc if wc else 0, c['SQ_WAIT_ANY'] / wc if wc else 0, c['SQ_WAIT_INST_LDS'] / wc if wc else 0, gui / max(1.0, tot_ns)))
P
rm -rf $OUT/pmc_sq_a3 $OUT/pmc_sq_a4 $OUT/pmc_g_a3 $OUT/pmc_g_a4 $OUT/st_a3 $OUT/st_a4
