#!/bin/bash
# round 6, call N: flakiness and breadth -- the GPU suite twice more (the driver runs it with -x), a wider fuzz with other seeds, the drop-in loop on a2 / a3
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06n
mkdir -p $OUT
for i in 1 2; do python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed" | tail -1; done > $OUT/pytest_gpu_repeats.txt 2>&1; cat $OUT/pytest_gpu_repeats.txt
FUZZ_N=24 FUZZ_KEYS=a2,a4,a3,p2,dn_lite5,dn_lite10,lite2,lite4,lite8,l25 FUZZ_SEED=61 FUZZ_CROPS=12 timeout 1500 python tools/fuzz_gpu.py > $OUT/fuzz_wide.txt 2>&1; echo "fuzz rc=$?"; grep -v amdgpu $OUT/fuzz_wide.txt | tail -30
