#!/bin/bash
# Same-box A/B of library variants moephoto_amd/_abl/lib_{old,new}.so (box-to-box spread is +-5 %, larger than most kernel tweaks):
# per-layer wall times twice, then s_memtime traces of the trunk convs at B=48.  Run through gpurun.
cp moephoto_amd/libmoephoto_amd.so /tmp/lib_orig.so
for rep in 1 2; do for f in old new; do
  cp moephoto_amd/_abl/lib_$f.so moephoto_amd/libmoephoto_amd.so
  echo "== $f"; python tools/gpu_diag.py layers 2>&1 | grep -E "B=12 layers \*(c1_|c2_|up0|up1)|B=12 whole"
done; done
for f in old new; do cp moephoto_amd/_abl/lib_$f.so moephoto_amd/libmoephoto_amd.so; echo "== trace $f"; for k in c1_3 c2_3 convt_R1.up0 convt_R1.up1; do MOE_TRACE_KEY=$k MOE_DBG=64 PROF_ITER=1 PROF_B=48 python tools/prof_workload.py >/dev/null 2>&1; python tools/show_trace_sp.py | head -1; done; done
cp /tmp/lib_orig.so moephoto_amd/libmoephoto_amd.so
