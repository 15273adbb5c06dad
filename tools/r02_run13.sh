#!/bin/bash
mkdir -p gpurun_out/r02l
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/micro/buffer_oor.hip -o /tmp/buffer_oor 2>/dev/null && timeout 60 /tmp/buffer_oor > gpurun_out/r02l/buffer_oor.txt 2>&1
cat gpurun_out/r02l/buffer_oor.txt
