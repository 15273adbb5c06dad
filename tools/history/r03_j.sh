#!/bin/bash
# A/B of two library builds on the headline bench and on a2 / dn_lite5 frames (same box)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
A=${AB_A:-x3old}; B=${AB_B:-x3new}
bash tools/ab_libs.sh $A moephoto_amd/_abl/lib_$A.so $B moephoto_amd/_abl/lib_$B.so
cp moephoto_amd/libmoephoto_amd.so /tmp/lib_product2.so
for rep in 1 2; do for t in $A $B; do
  cp moephoto_amd/_abl/lib_$t.so moephoto_amd/libmoephoto_amd.so
  for m in "SR a2" "DN lite5"; do echo -n "$t "; TM_ONLY="$m" TM_PREC=auto timeout 300 python tools/time_models.py 2>/dev/null | tail -1; done
done; done
cp /tmp/lib_product2.so moephoto_amd/libmoephoto_amd.so
