#!/bin/bash
# Ablation variants of conv3x3_ps4 (moephoto_amd/_abl/lib_<tag>.so from tools/mk_variant.sh) inside one call: frame time, the R / U up-conv groups, shader clock
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${R04_TAG:-r04c}
mkdir -p $OUT
cp moephoto_amd/libmoephoto_amd.so /tmp/lib_product.so
run() {
  cp "$2" moephoto_amd/libmoephoto_amd.so
  timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --sustain 3 --no-noise-input --no-dropin-loop 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
ks={k['layer_key']:k for k in r.get('roofline_kernels',[])}
c=r.get('clock',{})
print('%-8s %.3f ms/frame | R %.3f ms %.3f | U %.3f ms %.3f | arsb %.3f | sclk %s GHz %s W' % ('$1', r['ms_per_step'], ks['convt_R1.up1']['ms_per_frame'], ks['convt_R1.up1']['frac'], ks['u.up1']['ms_per_frame'], ks['u.up1']['frac'], ks['arsb']['ms_per_frame'], c.get('sclk_ghz_mean'), c.get('power_w_mean')))"
}
for rep in 1 2; do
  run product /tmp/lib_product.so
  for t in "$@"; do run $t moephoto_amd/_abl/lib_$t.so; done
done 2>&1 | tee $OUT/ablation.txt
cp /tmp/lib_product.so moephoto_amd/libmoephoto_amd.so
