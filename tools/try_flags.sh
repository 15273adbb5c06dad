#!/bin/bash
# usage: tools/try_flags.sh "<hipcc -D flags>"   -- rebuild with experimental flags, time the Net4x layers on the GPU box
cd "$(dirname "$0")/.."
MOE_HIPCC_FLAGS="$1" python -m moephoto_amd.build --force 2>&1 | grep -E "error|warning"
/usr/local/graft/bin/gpurun --timeout 600 -- 'timeout 120 python tools/gpu_diag.py layers 2>&1 | grep -E "B=12 layers \*(c1_|up1)|B=12 whole"' 2>&1 | grep -E "B=12|left"
