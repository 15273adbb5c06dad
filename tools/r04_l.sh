#!/bin/bash
# SQ / GRBM counters of the bench's launches with the trunk on arsb_s (and on arsb32c), kernel include regex: arsb
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${R04_TAG:-r04l}
mkdir -p $OUT
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain 0 --no-noise-input --no-dropin-loop --no-extras"
for impl in s v3; do
  MOE_ARSB_IMPL=$impl timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT --kernel-include-regex "arsb" -d $OUT/pmc_sq_$impl -o pmc -f csv -- $CMD > $OUT/sq_$impl.log 2>&1
  MOE_ARSB_IMPL=$impl timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-include-regex "arsb" -d $OUT/pmc_grbm_$impl -o pmc -f csv -- $CMD > $OUT/grbm_$impl.log 2>&1
  python - $OUT $impl <<'PY'
import csv, glob, sys, os
from collections import defaultdict
out, impl = sys.argv[1], sys.argv[2]
K = defaultdict(float); D = 0.0; n = 0
for f in glob.glob(os.path.join(out, 'pmc_sq_' + impl, '**', '*counter_collection.csv'), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        K[r['Counter_Name']] += float(r['Counter_Value'] or 0)
G = 0.0; DG = 0.0
for f in glob.glob(os.path.join(out, 'pmc_grbm_' + impl, '**', '*counter_collection.csv'), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        G += float(r['Counter_Value'] or 0)
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id']); DG += float(r['End_Timestamp']) - float(r['Start_Timestamp']); n += 1
w = K['SQ_WAVE_CYCLES']
print('%-3s dispatches %d, %.3f ms each | MFMA busy %.4f | wait_any %.4f | wait_inst_any %.4f | active_inst %.4f | clock %.3f GHz | MFMA insts %.3e' % (impl, n, DG / max(1, n) / 1e6,
      K['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * w), K['SQ_WAIT_ANY'] / w, K['SQ_WAIT_INST_ANY'] / w, K['SQ_ACTIVE_INST_ANY'] / w, G / 8 / max(1, DG), K['SQ_INSTS_MFMA']))
PY
done
