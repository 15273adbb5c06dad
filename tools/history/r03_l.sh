#!/bin/bash
# SEDN (l25): tests of the fused block tail, then 1080p frame times with two library builds alternating
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
A=${AB_A:-spold}; B=${AB_B:-spnew}
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "sedn or l25 or golden" 2>&1 | tail -2
cp moephoto_amd/libmoephoto_amd.so /tmp/lib_product3.so
for rep in 1 2; do for t in $A $B; do
  cp moephoto_amd/_abl/lib_$t.so moephoto_amd/libmoephoto_amd.so
  echo -n "$t "; TM_ONLY="DN l25" TM_PREC=auto timeout 300 python tools/time_models.py 2>/dev/null | tail -1
done; done
cp /tmp/lib_product3.so moephoto_amd/libmoephoto_amd.so
