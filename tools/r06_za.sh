#!/bin/bash
# round 6, call ZA: conv64_x3 with the wave of the output channels 48..63 idle on the 48-channel nets -- parity (lite, dn_lite, a2), A/B against the build without
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06za
mkdir -p $OUT
timeout 1200 python -m pytest tests -q -m gpu -x -k "lite or golden or kernel_forms or stub or split_operand or layer_by_layer or exact" 2>&1 | tail -6 > $OUT/pytest.txt; cat $OUT/pytest.txt
cp moephoto_amd/libmoephoto_amd.so /tmp/lib_product.so
{
for rep in 1 2; do for v in product x3_noidle; do
  [ $v = product ] && cp /tmp/lib_product.so moephoto_amd/libmoephoto_amd.so || cp moephoto_amd/_abl/lib_$v.so moephoto_amd/libmoephoto_amd.so
  echo "== $v"; TM_PREC=auto timeout 400 python tools/time_models.py 2>&1 | grep -E "lite2|lite4|lite5"
done; done
cp /tmp/lib_product.so moephoto_amd/libmoephoto_amd.so
} > $OUT/ab_idle.txt 2>&1; cat $OUT/ab_idle.txt
