#!/bin/bash
# r04n: I/O-edge kernels (test + bench extras), per-kernel stats of lite2 / dn_lite5 / a2 frames
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${R04_TAG:-r04n}
mkdir -p $OUT
timeout 300 python -m pytest tests -m gpu -x -q -k "io_edges or e2e_uint8 or sixteen or 16" 2>&1 | grep -v amdgpu.ids | tail -5 > $OUT/pytest_io.txt
python - > $OUT/io_edges.txt 2>&1 <<'P'
import sys, json
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, bench_extra
from moephoto_amd import _lib
print(json.dumps(bench_extra.io_edges(torch, _lib, torch.device('cuda:0')), indent=1))
P
for m in "SR lite2" "DN lite5" "SR a2"; do
  tag=$(echo $m | tr ' ' '_')
  TM_ONLY="$m" TM_PREC=auto timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/st_$tag -o st -f csv -- python tools/time_models.py > $OUT/tm_$tag.txt 2>&1
  f=$(find $OUT/st_$tag -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $OUT/kernel_stats_$tag.csv
  rm -rf $OUT/st_$tag
done
tail -3 $OUT/pytest_io.txt; cat $OUT/io_edges.txt | grep -E "frac|ms\"" ; for f in $OUT/kernel_stats_*.csv; do echo $f; cut -d, -f1-5 $f | cut -c1-150 | head -14; done
