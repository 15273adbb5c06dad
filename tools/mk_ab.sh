#!/bin/bash
# Build moephoto_amd/_abl/lib_old.so (conv3x3_sp.hip from git HEAD, or the revision given as $1) and lib_new.so (working tree)
# for tools/ab_layers.sh.  Run here (no GPU needed); the other objects come from moephoto_amd/_obj (python -m moephoto_amd.build first).
set -e
cd "$(dirname "$0")/.."
REV=${1:-HEAD}
mkdir -p moephoto_amd/_abl /tmp/t
rm -f moephoto_amd/_abl/lib_old.so moephoto_amd/_abl/lib_new.so
git show $REV:moephoto_amd/csrc/conv3x3_sp.hip > moephoto_amd/csrc/_old_sp.hip
OBJS="moephoto_amd/_obj/conv_mfma.o moephoto_amd/_obj/conv3x3_pp.o moephoto_amd/_obj/misc_kernels.o moephoto_amd/_obj/engine.o moephoto_amd/_obj/planner.o"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c moephoto_amd/csrc/_old_sp.hip -o /tmp/t/old_sp.o
rm moephoto_amd/csrc/_old_sp.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $MOE_HIPCC_FLAGS -c moephoto_amd/csrc/conv3x3_sp.hip -o /tmp/t/new_sp.o
hipcc --offload-arch=gfx950 -shared -fPIC -o moephoto_amd/_abl/lib_old.so $OBJS /tmp/t/old_sp.o
hipcc --offload-arch=gfx950 -shared -fPIC -o moephoto_amd/_abl/lib_new.so $OBJS /tmp/t/new_sp.o
echo built old=$REV new=worktree
