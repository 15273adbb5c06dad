#!/bin/bash
mkdir -p gpurun_out/r02f
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "split_operand or fused_arsb or resize or p2 or net_forward_vs" > gpurun_out/r02f/pytest.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r02f/pytest.log | cut -c1-300
timeout 600 python tools/diag_arsb.py > gpurun_out/r02f/diag_arsb.txt 2>&1; echo "diag rc=$?"
grep -E "^a4 B=12" gpurun_out/r02f/diag_arsb.txt
timeout 600 python bench.py --no-noise-input --cpu-tiles 2 --sustain 3 > gpurun_out/r02f/bench.json 2> gpurun_out/r02f/bench.err; echo "bench rc=$?"
cut -c1-400 gpurun_out/r02f/bench.json
