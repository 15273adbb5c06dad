#!/bin/bash
# Which kernel instantiations does each (zoo key, precision, shape class) resolve to?  One short process per case under rocprofv3 --kernel-trace --stats, condensed by
# tools/kernel_table.py into tests/golden/kernel_resolution.json (held against the library's compiled instantiations by a CPU test: tests/test_host.py).
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/ktable
rm -rf $OUT; mkdir -p $OUT
for key in a2 a3 a4 dn_lite5 l25 lite2 lite4 lite8; do for prec in auto fp16 fp16x3; do for cls in frame tile odd; do
  if [ $prec != auto ] && [ $cls != frame ]; then continue; fi      # (the non-default arithmetics: the batched frame only)
  d=$OUT/${key}__${prec}__${cls}
  timeout 120 rocprofv3 --kernel-trace --stats -d $d -o t -f csv -- python tools/kernel_table_case.py $key $prec $cls > $d.log 2>&1 || echo "FAILED $key $prec $cls: $(tail -1 $d.log)"
  f=$(find $d -name '*kernel_stats.csv' 2>/dev/null | head -1); [ -n "$f" ] && cp $f $d.csv; rm -rf $d
done; done; done
python tools/kernel_table.py $OUT > $OUT/kernel_resolution.json; python -c "import json; d=json.load(open('$OUT/kernel_resolution.json')); print(len(d['cases']), 'cases;', len(d['launched']), 'instantiations launched of', len(d['compiled']))"
