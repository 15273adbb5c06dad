#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "golden or kernel_forms or config5" > gpurun_out/pytest26.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest26.log
bash tools/ab_libs.sh new moephoto_amd/_abl/lib_after_epi7.so old moephoto_amd/_abl/lib_before_epi7.so
