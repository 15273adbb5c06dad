#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06j
mkdir -p $OUT
MOE_DIST_EXCHANGE=p2p MOE_FORCE_DEVICE=0 MOE_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python bench.py --gpus 3 --steps 2 --warmup 1 --no-cpu-baseline --sustain 0 --no-noise-input > $OUT/g3_p2p.out 2> $OUT/g3_p2p.err; echo "rc=$?"
grep -n "Error\|error\|rank1\|rank2" $OUT/g3_p2p.err | head -40
MOE_FORCE_DEVICE=0 MOE_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python bench.py --gpus 3 --steps 2 --warmup 1 --no-cpu-baseline --sustain 0 --no-noise-input > $OUT/g3_a2a.out 2> $OUT/g3_a2a.err; echo "a2a rc=$?"
tail -c 600 $OUT/g3_a2a.out
