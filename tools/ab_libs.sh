#!/bin/bash
# A/B of library builds inside ONE gpurun call (boxes differ by several per cent): tools/ab_libs.sh <tag> <lib.so> [<tag> <lib.so> ...]
# runs the short benchmark with each library copied over the product one, twice, interleaved.
cd "$(dirname "$0")/.."
cp moephoto_amd/libmoephoto_amd.so /tmp/lib_product.so
run() {
  cp "$2" moephoto_amd/libmoephoto_amd.so
  timeout 300 python bench.py --steps ${AB_STEPS:-10} --warmup 3 --no-cpu-baseline --sustain 0 --no-noise-input --no-dropin-loop 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'ms_per_step', d['ms_per_step'], [(k['layer_key'], k['ms_per_frame']) for k in d.get('roofline_kernels', [])])"
}
for rep in 1 2; do
  args=("$@")
  while [ ${#args[@]} -ge 2 ]; do run "${args[0]}" "${args[1]}"; args=("${args[@]:2}"); done
done
cp /tmp/lib_product.so moephoto_amd/libmoephoto_amd.so
