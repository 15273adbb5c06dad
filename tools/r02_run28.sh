#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "golden or kernel_forms or config5 or fuzz or whole" > gpurun_out/pytest28.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest28.log
bash tools/ab_libs.sh drain2 moephoto_amd/_abl/lib_drain1.so drain1 moephoto_amd/_abl/lib_drain0.so
