#!/bin/bash
# counters of the kernels matching $PMC_RE on one model's 1080p frame (tools/time_models.py): MFMA busy, instruction mix per MFMA, waits, effective clock
#   PMC_TAG=r06z3 PMC_MODEL="SR lite8" PMC_RE="conv1x1" bash tools/pmc_kernels.sh
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${PMC_TAG:-pmc}
mkdir -p $OUT
M="${PMC_MODEL:-SR lite8}"; RE="${PMC_RE:-conv1x1}"
TM_ONLY="$M" TM_PREC=auto timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY --kernel-include-regex "$RE" -d $OUT/pmc_a -o pmc -f csv -- python tools/time_models.py > $OUT/pmc_a.log 2>&1
TM_ONLY="$M" TM_PREC=auto timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM --kernel-include-regex "$RE" -d $OUT/pmc_b -o pmc -f csv -- python tools/time_models.py > $OUT/pmc_b.log 2>&1
TM_ONLY="$M" TM_PREC=auto timeout 300 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_WAIT_INST_VMEM --kernel-include-regex "$RE" -d $OUT/pmc_c -o pmc -f csv -- python tools/time_models.py > $OUT/pmc_c.log 2>&1
TM_ONLY="$M" TM_PREC=auto timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/st -o st -f csv -- python tools/time_models.py > $OUT/st.log 2>&1
python - $OUT "$RE" <<'P'
import csv, glob, sys, collections, re
out, rex = sys.argv[1], sys.argv[2]
def key(n):
    m = re.search(r'((?:%s)\w*<[^>]*>|(?:%s)\w*)' % (rex, rex), n)
    return m.group(1) if m else None
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = key(r['Kernel_Name'])
        if k: agg[k][r['Counter_Name']] += float(r['Counter_Value'])
dur = {}
for f in glob.glob(out + '/st/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = key(r['Name'])
        if k: dur[k] = (float(r['TotalDurationNs']), int(r['Calls']))
for k, c in sorted(agg.items()):
    wc = max(1.0, c['SQ_WAVE_CYCLES']); mf = max(1.0, c['SQ_INSTS_MFMA']); tot_ns, calls = dur.get(k, (0.0, 1))
    print('%-40s launches %3d total %8.2f ms | per wave cycle: MFMA busy/4 %.3f  VALU active %.3f  LDS active %.3f  VMEM active %.3f  issuing any %.3f | wait-inst any %.3f  LDS %.3f  VMEM %.3f  wait-any %.3f | per MFMA: wave cycles %.1f  VALU %.2f  LDS %.2f  VMEM %.3f (rd %.3f wr %.3f)  SALU %.2f | clock %.2f GHz' % (
        k, calls, tot_ns / 1e6, c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * wc), c['SQ_ACTIVE_INST_VALU'] / wc, c['SQ_ACTIVE_INST_LDS'] / wc, c['SQ_ACTIVE_INST_VMEM'] / wc, c['SQ_ACTIVE_INST_ANY'] / wc,
        c['SQ_WAIT_INST_ANY'] / wc, c['SQ_WAIT_INST_LDS'] / wc, c['SQ_WAIT_INST_VMEM'] / wc, c['SQ_WAIT_ANY'] / wc,
        wc / mf, c['SQ_INSTS_VALU'] / mf, c['SQ_INSTS_LDS'] / mf, c['SQ_INSTS_VMEM'] / mf, c['SQ_INSTS_VMEM_RD'] / mf, c['SQ_INSTS_VMEM_WR'] / mf, c['SQ_INSTS_SALU'] / mf, c['GRBM_GUI_ACTIVE'] / 8.0 / max(1.0, tot_ns)))
P
rm -rf $OUT/pmc_a $OUT/pmc_b $OUT/pmc_c $OUT/st
