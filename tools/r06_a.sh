#!/bin/bash
# round 6, call A: the Winograd probe beside conv3x3_ps4<1> on the same box, PMC passes over the probe; the GPU suite's new / changed tests
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06a
mkdir -p $OUT
P=tools/micro/bin/wino_probe
{
echo "== wino_probe"; timeout 300 $P 96 512 512 20
echo "== wino_probe again (clock settled)"; timeout 300 $P 96 512 512 60 | tail -2
echo "== conv3x3_ps4<1> / <2> looped alone on this box (tools/kernel_power.py)"; timeout 300 python tools/kernel_power.py 3 u.up1,convt_R1.up1 2>&1 | grep -v amdgpu.ids
} > $OUT/wino_probe.txt 2>&1
cat $OUT/wino_probe.txt
for pass in "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" "sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "grbm GRBM_GUI_ACTIVE" "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $pass; name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex wino_kernel -d $OUT/pmc_$name -o pmc -f csv -- $P 96 512 512 3 > $OUT/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
done
python - <<'PY' > $OUT/wino_pmc.txt 2>&1
import csv, glob, collections
K = collections.defaultdict(float); D = {}
for f in glob.glob('gpurun_out/r06a/pmc_*/**/*counter_collection.csv', recursive=True):
    name = f.split('pmc_')[1].split('/')[0]
    for r in csv.DictReader(open(f)):
        if 'ILi0E' not in r['Kernel_Name'] and '<0>' not in r['Kernel_Name']: continue
        K[r['Counter_Name']] += float(r['Counter_Value'] or 0)
        D.setdefault(name, {})[r['Dispatch_Id']] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
n = {k: len(v) for k, v in D.items()}
print('dispatches per pass', n)
for k in sorted(K): print('%-28s %.4g' % (k, K[k]))
if K.get('SQ_WAVE_CYCLES'):
    print('MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_WAVE_CYCLES) = %.3f' % (K['SQ_VALU_MFMA_BUSY_CYCLES'] / 4 / K['SQ_WAVE_CYCLES']))
    print('per MFMA: wave quad-cycles %.2f, wait_any %.2f, wait_inst_any %.2f, active_inst_any %.2f' % tuple(K[c] / K['SQ_INSTS_MFMA'] for c in ('SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY')))
    print('LDS bank conflict cycles per dispatch %.3g' % (K['SQ_LDS_BANK_CONFLICT'] / n['sq']))
if K.get('SQ_INSTS_VALU'):
    m = K['SQ_INSTS_MFMA'] / n['sq'] * n['sq2']
    print('per MFMA: VALU (incl. MFMA) %.2f, SALU %.2f, LDS %.2f, VMEM rd %.3f; wait_inst_lds quad-cycles %.2f, lds_idx_active %.2f' % tuple(K[c] / m for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_WAIT_INST_LDS', 'SQ_LDS_IDX_ACTIVE')))
if K.get('GRBM_GUI_ACTIVE'):
    dur = sum(D['grbm'].values())
    print('effective clock = GRBM_GUI_ACTIVE / 8 / duration = %.3f GHz; %.3f ms per dispatch (profiled)' % (K['GRBM_GUI_ACTIVE'] / 8 / dur, dur / n['grbm'] / 1e6))
if K.get('FETCH_SIZE'): print('fetch %.3f GB per dispatch (x2 corrected), write %.3f GB' % (K['FETCH_SIZE'] * 1024 * 2 / n['fetch'] / 1e9, K.get('WRITE_SIZE', 0) * 1024 / max(1, n.get('write', 1)) / 1e9))
PY
cat $OUT/wino_pmc.txt
python -m pytest tests -x -q -m gpu -k "blend_tile or calibrate or small_launch or integration_md or dropin" 2>&1 | tail -5 > $OUT/pytest_subset.txt; cat $OUT/pytest_subset.txt
