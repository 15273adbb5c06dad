#!/bin/bash
mkdir -p gpurun_out/r02v
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "kernel_forms or golden" > gpurun_out/r02v/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02v/pytest.log
FUZZ_N=8 FUZZ_SEED=7 timeout 900 python tools/fuzz_gpu.py > gpurun_out/r02v/fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -16 gpurun_out/r02v/fuzz.txt
MOE_SP_IMPL=rw FUZZ_N=6 FUZZ_SEED=11 FUZZ_KEYS=a2,a3,a4 timeout 900 python tools/fuzz_gpu.py > gpurun_out/r02v/fuzz_rw.txt 2>&1; echo "fuzz rw rc=$?"; tail -7 gpurun_out/r02v/fuzz_rw.txt
