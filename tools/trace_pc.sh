#!/bin/bash
# Build a -DPC_TRACE variant of the library (arsb_pc.hip with s_memtime stamps around every step barrier) into
# moephoto_amd/_abl/lib_pc_trace.so (run HERE, no GPU needed); on the GPU box tools/show_trace_pc.py prints the per-step cycle table.
set -e
cd "$(dirname "$0")/.."
mkdir -p moephoto_amd/_abl /tmp/t
OBJS=$(ls moephoto_amd/_obj/*.o | grep -v arsb_pc)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DPC_TRACE=${PC_TRACE:-1} ${PC_DEFS:-} -c moephoto_amd/csrc/arsb_pc.hip -o /tmp/t/arsb_pc_trace.o
hipcc --offload-arch=gfx950 -shared -fPIC -o moephoto_amd/_abl/lib_pc_trace${PC_TAG:-}.so $OBJS /tmp/t/arsb_pc_trace.o
echo built moephoto_amd/_abl/lib_pc_trace${PC_TAG:-}.so
