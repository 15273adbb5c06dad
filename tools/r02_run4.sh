#!/bin/bash
mkdir -p gpurun_out/r02d
timeout 600 python tools/diag_arsb.py > gpurun_out/r02d/diag_arsb.txt 2>&1; echo "diag rc=$?"
grep -E "out err|^a4 B=12" gpurun_out/r02d/diag_arsb.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "fused_arsb" > gpurun_out/r02d/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r02d/pytest.log
