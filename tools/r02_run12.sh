#!/bin/bash
mkdir -p gpurun_out/r02k
MOE_ARSB_IMPL=pc timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused_arsb or net_forward or config5 or batches" > gpurun_out/r02k/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02k/pytest.log
timeout 300 python tools/diag_arsb.py > gpurun_out/r02k/diag_arsb.txt 2>&1; echo "diag rc=$?"; tail -4 gpurun_out/r02k/diag_arsb.txt
timeout 300 python tools/show_trace_pc.py > gpurun_out/r02k/trace_pc.txt 2>&1; echo "trace rc=$?"
cat gpurun_out/r02k/trace_pc.txt | cut -c1-400 | head -12
