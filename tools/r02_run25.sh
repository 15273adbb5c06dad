#!/bin/bash
mkdir -p gpurun_out/r02w
export TMPDIR=/tmp
for case in "DN l25:auto" "DN lite5:auto" "SR a3:auto"; do
  name=${case%%:*}; prec=${case##*:}; tag=$(echo "$name-$prec" | tr ' ' '_')
  TM_ONLY="$name" TM_PREC=$prec timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r02w/$tag -o tm -f csv -- python tools/time_models.py > gpurun_out/r02w/$tag.log 2>&1
  echo "== $name $prec rc=$?"; grep "ms/frame" gpurun_out/r02w/$tag.log
  f=$(find gpurun_out/r02w/$tag -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:12]:
    print('  %6.2f%% %6d calls avg %9.1f us  %s'%(100*float(r['TotalDurationNs'])/tot,int(r['Calls']),float(r['AverageNs'])/1e3,r['Name'][:110]))
PY
done
