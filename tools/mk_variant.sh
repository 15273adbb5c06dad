#!/bin/bash
# Build moephoto_amd/_abl/lib_<tag>.so: the product library with ONE source recompiled with extra -D flags (kernel A/B inside one gpurun call,
# see tools/ab_libs.sh).   tools/mk_variant.sh <tag> <source.hip> [-DFOO=1 ...]      (python -m moephoto_amd.build first)
set -e
cd "$(dirname "$0")/.."
TAG=$1; SRC=$2; shift; shift
BASE=$(basename $SRC .hip)
mkdir -p moephoto_amd/_abl /tmp/t
EXTRA=$(python - <<P
from moephoto_amd import build
print(' '.join(build.EXTRA_FLAGS.get('$BASE.hip', [])))
P
)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -c moephoto_amd/csrc/$BASE.hip -o /tmp/t/${BASE}_$TAG.o $EXTRA "$@"
OBJS=$(ls moephoto_amd/_obj/*.o | grep -v "/$BASE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o moephoto_amd/_abl/lib_$TAG.so $OBJS /tmp/t/${BASE}_$TAG.o
echo built moephoto_amd/_abl/lib_$TAG.so
