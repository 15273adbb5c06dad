#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/pytest29.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest29.log
bash tools/ab_libs.sh flag moephoto_amd/_abl/lib_flag.so noflag moephoto_amd/_abl/lib_noflag.so
