#!/bin/bash
cp moephoto_amd/libmoephoto_amd.so /tmp/lib_orig.so
for n in steps min; do
  cp moephoto_amd/_abl/lib_sptrace_$n.so moephoto_amd/libmoephoto_amd.so
  for k in u.up1 convt_R1.up1; do echo "== $n key $k"; MOE_TRACE_KEY=$k MOE_DBG=64 PROF_ITER=1 PROF_B=12 timeout 120 python tools/prof_workload.py > /dev/null 2>&1; python tools/show_trace_sp.py | head -6; done
done
cp /tmp/lib_orig.so moephoto_amd/libmoephoto_amd.so
