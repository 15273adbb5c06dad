#!/bin/bash
# conv64_sq variants (moephoto_amd/_abl/lib_sq_<tag>.so from tools/mk_variant.sh) inside one call: a2 launch set, the split-operand layers' times
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${R04_TAG:-r04q}
mkdir -p $OUT
cp moephoto_amd/libmoephoto_amd.so /tmp/lib_product.so
for rep in 1 2; do
  SQ_TAG=q8 SQ_IMPLS=p timeout 120 python tools/time_sq.py 2>/dev/null | grep "launch set"
  SQ_TAG=product SQ_IMPLS=s timeout 120 python tools/time_sq.py 2>/dev/null | grep "launch set"
  for t in "$@"; do
    cp moephoto_amd/_abl/lib_sq_$t.so moephoto_amd/libmoephoto_amd.so
    SQ_TAG=$t SQ_IMPLS=s timeout 120 python tools/time_sq.py 2>/dev/null | grep "launch set"
    cp /tmp/lib_product.so moephoto_amd/libmoephoto_amd.so
  done
done 2>&1 | tee $OUT/variants.txt
cp /tmp/lib_product.so moephoto_amd/libmoephoto_amd.so
