#!/bin/bash
mkdir -p gpurun_out/r02o
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "lite or fuzz or golden or drop_in" > gpurun_out/r02o/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02o/pytest.log
for v in 1 0; do
echo "== MOE_CONV1X1=$v"
MOE_CONV1X1=$v TM_ONLY="SR lite2" TM_PREC=auto,fp16 timeout 300 python tools/time_models.py 2>&1 | grep ms/frame
MOE_CONV1X1=$v TM_ONLY="SR lite4" TM_PREC=auto,fp16 timeout 300 python tools/time_models.py 2>&1 | grep ms/frame
done
