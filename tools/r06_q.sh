#!/bin/bash
# round 6, call Q: the GPU suite on the tree with conv3x3_ps9 and the zero-block skip in conv3x3_ps4 / ps9; A/B of the skip on the headline bench (same box, interleaved)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06q
mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
AB_STEPS=10 bash tools/ab_libs.sh zskip moephoto_amd/libmoephoto_amd.so nozskip moephoto_amd/_abl/lib_ps4_nozskip.so > $OUT/ab_ps4_zskip.txt 2>&1; cat $OUT/ab_ps4_zskip.txt
timeout 600 python tools/time_models.py 2>&1 | grep -v amdgpu.ids | grep auto > $OUT/time_models.txt; cat $OUT/time_models.txt
