#!/bin/bash
mkdir -p gpurun_out/r02s
export TMPDIR=/tmp
for impl in rw sp; do
MOE_SP_IMPL=$impl timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r02s/$impl -o b -f csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --sustain 0 --no-noise-input > gpurun_out/r02s/$impl.log 2>&1
f=$(find gpurun_out/r02s/$impl -name "*kernel_stats.csv" | head -1)
echo "== $impl"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:9]:
    print('  %6.2f%% %6d calls avg %9.1f us  %s'%(100*float(r['TotalDurationNs'])/tot,int(r['Calls']),float(r['AverageNs'])/1e3,r['Name'][:90]))
PY
done
