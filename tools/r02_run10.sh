#!/bin/bash
mkdir -p gpurun_out/r02i
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "split_operand or fused_arsb or net_forward_vs or docrop or ragged" > gpurun_out/r02i/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r02i/pytest.log | cut -c1-300
timeout 600 python tools/diag_arsb.py > gpurun_out/r02i/diag_arsb.txt 2>&1; echo "diag rc=$?"
grep -E "^a4 B=12" gpurun_out/r02i/diag_arsb.txt
python tools/show_trace_arsb.py lib_trace.so 2>&1 | grep "wave 0"
timeout 600 python bench.py --no-noise-input --cpu-tiles 2 --sustain 3 > gpurun_out/r02i/bench.json 2> gpurun_out/r02i/bench.err; echo "bench rc=$?"
cut -c1-260 gpurun_out/r02i/bench.json
