#!/bin/bash
# ablations of conv64_q8 on the a2 frame (tools/mk_variant.sh q8<tag> conv64_q8.hip -D...): which part of a patch costs what
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
cp moephoto_amd/libmoephoto_amd.so /tmp/prod.so
run() { cp $2 moephoto_amd/libmoephoto_amd.so; echo -n "$1  "; TM_ONLY="SR a2" TM_PREC=auto timeout 200 python tools/time_models.py 2>/dev/null | tail -1; }
run product /tmp/prod.so
for v in dbg1 dbg2 dbg3 nocvt nolo8 nost; do run $v moephoto_amd/_abl/lib_q8$v.so; done
run product /tmp/prod.so
cp /tmp/prod.so moephoto_amd/libmoephoto_amd.so
