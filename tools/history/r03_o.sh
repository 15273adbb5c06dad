#!/bin/bash
# launch-set size (option tiles_per_batch) against frame time: do the exact layers' streams fit the 256-MiB Infinity Cache when the sets are smaller?
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for m in "SR a2"; do
  for t in ${TPB:-2 3 4 6 8 12 16 24}; do
    echo -n "tiles_per_batch $t  "; MOE_TILES_PER_BATCH=$t TM_ONLY="$m" TM_PREC=auto timeout 300 python tools/time_models.py 2>/dev/null | tail -1
  done
done
