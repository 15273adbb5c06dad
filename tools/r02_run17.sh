#!/bin/bash
for v in 1; do echo "== MOE_CONV1X1=$v"; for k in lite4 lite8 lite2; do MOE_CONV1X1=$v timeout 200 python tools/diag_lite.py $k 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-200; done; done
mkdir -p gpurun_out/r02o
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "lite or fuzz or golden or drop_in or fused_arsb" > gpurun_out/r02o/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02o/pytest.log
MOE_ARSB_IMPL=pc timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused_arsb or net_forward" > gpurun_out/r02o/pytest_pc.log 2>&1; echo "pytest pc rc=$?"; tail -2 gpurun_out/r02o/pytest_pc.log
for m in "SR lite2" "SR lite4"; do TM_ONLY="$m" TM_PREC=auto,fp16 timeout 300 python tools/time_models.py 2>&1 | grep ms/frame; done
