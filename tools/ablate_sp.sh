#!/bin/bash
# Compile-time timing ablations of conv3x3_sp_kernel: builds variants of the library under moephoto_amd/_abl/ (run here),
# then `tools/ablate_sp.sh run` on the GPU box times the fixed Net4x workload with each (results are WRONG by design).
set -u
cd "$(dirname "$0")/.."
if [ "${1:-build}" = build ]; then
  mkdir -p moephoto_amd/_abl
  for v in "${@:2}"; do
    name=${v%%=*}; flags=${v#*=}
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c moephoto_amd/csrc/conv3x3_sp.hip -o moephoto_amd/_abl/sp_$name.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o moephoto_amd/_abl/lib_$name.so moephoto_amd/_obj/conv_mfma.o moephoto_amd/_obj/conv3x3_pp.o \
      moephoto_amd/_abl/sp_$name.o moephoto_amd/_obj/misc_kernels.o moephoto_amd/_obj/engine.o moephoto_amd/_obj/planner.o && echo built $name
    rm -f moephoto_amd/_abl/sp_$name.o
  done
else
  cp moephoto_amd/libmoephoto_amd.so /tmp/lib_orig.so
  for f in moephoto_amd/_abl/lib_*.so; do
    cp $f moephoto_amd/libmoephoto_amd.so
    echo "== $(basename $f)"
    if [ "$1" = trace ]; then   # cycle counts (s_memtime) are immune to the DVFS effect of the ablated data
      for k in ${TRACE_KEYS:-convt_R1.up1}; do echo "  key $k"; MOE_TRACE_KEY=$k MOE_DBG=64 PROF_ITER=1 PROF_B=${TRACE_B:-12} python tools/prof_workload.py > /dev/null 2>&1; python tools/show_trace_sp.py | head -1; done
    else
      python tools/gpu_diag.py layers 2>&1 | grep -E "B=12 layers \*(c1_|up0|up1)"
    fi
  done
  cp /tmp/lib_orig.so moephoto_amd/libmoephoto_amd.so
fi
