#!/bin/bash
# round-2 GPU call 1: bench (default arithmetic), the whole -m gpu suite, error / timing diagnostics
mkdir -p gpurun_out/r02a
python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r02a/bench.json
python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc=$?"
tail -40 gpurun_out/r02a/pytest.log
DIAG_KEYS=a2,a4,a3,dn_lite5 DIAG_PREC=auto,fp16 python tools/gpu_diag.py nets layers > gpurun_out/r02a/diag_stdout.log 2>&1
cp gpurun_out/diag.txt gpurun_out/r02a/diag.txt
TM_PREC=auto,fp16 python tools/time_models.py > gpurun_out/r02a/time_models.txt 2>&1
cat gpurun_out/r02a/time_models.txt
