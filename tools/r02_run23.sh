#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "golden or fused_arsb or split_operand or fuzz" > gpurun_out/pytest23.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest23.log
bash tools/ab_libs.sh new moephoto_amd/_abl/lib_new_waits.so old moephoto_amd/_abl/lib_before_waits.so
for l in new_waits before_waits; do cp moephoto_amd/_abl/lib_$l.so moephoto_amd/libmoephoto_amd.so; echo "== $l"; TM_ONLY="SR a2" TM_PREC=auto timeout 300 python tools/time_models.py 2>&1 | grep ms/frame; done
cp moephoto_amd/_abl/lib_new_waits.so moephoto_amd/libmoephoto_amd.so
