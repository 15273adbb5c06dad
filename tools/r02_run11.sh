#!/bin/bash
mkdir -p gpurun_out/r02j
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "fused_arsb or net_forward_vs" > gpurun_out/r02j/pytest.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/r02j/pytest.log | cut -c1-300
timeout 600 python tools/diag_arsb.py > gpurun_out/r02j/diag_arsb.txt 2>&1; echo "diag rc=$?"
grep -E "out err|tap arsb|impl|max " gpurun_out/r02j/diag_arsb.txt | head -60
