#!/bin/bash
# round-2 GPU call 2: fused ARSB kernel -- diagnostics, its test, bench
mkdir -p gpurun_out/r02b
timeout 600 python tools/diag_arsb.py > gpurun_out/r02b/diag_arsb.txt 2>&1; echo "diag rc=$?"
cat gpurun_out/r02b/diag_arsb.txt | tail -60
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "fused_arsb or config3 or net_forward_vs or docrop" > gpurun_out/r02b/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r02b/pytest.log
timeout 600 python bench.py --no-noise-input --cpu-tiles 2 --sustain 3 > gpurun_out/r02b/bench.json 2> gpurun_out/r02b/bench.err; echo "bench rc=$?"
cut -c1-1500 gpurun_out/r02b/bench.json
