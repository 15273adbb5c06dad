#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r05b
mkdir -p $OUT
cp moephoto_amd/libmoephoto_amd.so /tmp/lib_product.so
for rep in 1 2; do
for v in product nost nolost nodma noepi nofrag noall nolo; do
  if [ $v = product ]; then cp /tmp/lib_product.so moephoto_amd/libmoephoto_amd.so; else cp moephoto_amd/_abl/lib_a32_$v.so moephoto_amd/libmoephoto_amd.so; fi
  MOE_AUTO_CALIBRATE=0 timeout 120 python tools/kernel_power.py 3 arsb3 $v 2>&1 | grep -v amdgpu.ids | tail -4
done
done > $OUT/arsb32c_ablations2.txt 2>&1
cp /tmp/lib_product.so moephoto_amd/libmoephoto_amd.so
cat $OUT/arsb32c_ablations2.txt
