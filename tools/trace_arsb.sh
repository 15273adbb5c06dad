#!/bin/bash
# Build a -DARSB_TRACE variant of the library into moephoto_amd/_abl/lib_trace.so (run HERE, no GPU needed); on the GPU box
# tools/show_trace_arsb.py runs the fixed workload with it and prints the per-phase cycle table.
set -e
cd "$(dirname "$0")/.."
mkdir -p moephoto_amd/_abl /tmp/t
OBJS=$(ls moephoto_amd/_obj/*.o | grep -v arsb_fused)
for abl in ${ARSB_ABLS:-0}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -DARSB_TRACE -DARSB_ABL=$abl $MOE_HIPCC_FLAGS -c moephoto_amd/csrc/arsb_fused.hip -o /tmp/t/arsb_trace_$abl.o
  out=moephoto_amd/_abl/lib_trace$([ "$abl" = 0 ] || echo _$abl).so
  hipcc --offload-arch=gfx950 -shared -fPIC -o $out $OBJS /tmp/t/arsb_trace_$abl.o
  echo built $out
done
