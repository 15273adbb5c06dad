#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${R03_TAG:-r03d}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "${R03_K:-kernel_forms or forward_vs_reference or ragged or docrop_vs or full_size_properties or ensemble or fused_arsb or layer_by_layer or exact_blocks or run_plan_frames or config4}" > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_sel.log
cp moephoto_amd/libmoephoto_amd.so /tmp/lib_base.so
if ls moephoto_amd/_abl/lib_*.so > /dev/null 2>&1; then
  AB_STEPS=12 bash tools/ab_libs.sh base /tmp/lib_base.so $(for f in moephoto_amd/_abl/lib_*.so; do t=$(basename $f .so); echo ${t#lib_} $f; done) 2>&1 | tee $OUT/ab.txt
fi
for tpb in ${R03_TPB:-}; do
  python bench.py --steps 12 --warmup 3 --no-cpu-baseline --sustain 0 --no-noise-input --no-dropin-loop --tiles-per-batch $tpb 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiles-per-batch $tpb', d['ms_per_step'])"
done
