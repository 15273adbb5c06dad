#!/bin/bash
# A/B of library builds on one model's 1080p frame inside ONE gpurun call: TM_ONLY="SR a2" tools/ab_models.sh <tag> <lib.so> [<tag> <lib.so> ...]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
cp moephoto_amd/libmoephoto_amd.so /tmp/lib_product.so
for rep in 1 2; do
  args=("$@")
  while [ ${#args[@]} -ge 2 ]; do
    cp "${args[1]}" moephoto_amd/libmoephoto_amd.so
    echo -n "${args[0]}  "; TM_ONLY="${TM_ONLY:-SR a2}" TM_PREC=auto timeout 300 python tools/time_models.py 2>/dev/null | tail -1
    args=("${args[@]:2}")
  done
done
cp /tmp/lib_product.so moephoto_amd/libmoephoto_amd.so
