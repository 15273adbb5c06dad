#!/bin/bash
mkdir -p gpurun_out/r02p
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/r02p/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02p/pytest.log
timeout 300 python tools/diag_arsb.py 2>&1 | grep "x3-fuse\|arsb impl v1" 
for m in "SR a2" "SR lite2"; do TM_ONLY="$m" TM_PREC=auto timeout 300 python tools/time_models.py 2>&1 | grep ms/frame; done
