#!/bin/bash
# Builds a variant of libcmaxhip.so for a same-box A/B (tools/ab_builds.sh): the two kernel files recompiled with extra
# definitions, the host objects of the current build reused.   tools/build_variant.sh <name> "<-D...>"  ->  tools/ab/lib_<name>.so
set -e
NAME=$1; DEFS=$2
cd "$(dirname "$0")/../cmax_slam_amd/csrc"
make -s -j8 >/dev/null
mkdir -p ../../tools/ab build/var_$NAME
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function -Wno-unused-value"
for f in cmx_kernels cmx_binning; do /opt/rocm/bin/hipcc $FLAGS $DEFS -c -o build/var_$NAME/$f.o $f.hip & done
/opt/rocm/bin/hipcc $FLAGS $DEFS -x hip -c -o build/var_$NAME/cmx_pipeline.o cmx_pipeline.cpp &   # (launch-shape constants live there too)
wait
OBJS=$(ls build/*.o | grep -v -e cmx_kernels.o -e cmx_binning.o -e cmx_pipeline.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-soname,libcmaxhip.so -o ../../tools/ab/lib_$NAME.so build/var_$NAME/cmx_kernels.o build/var_$NAME/cmx_binning.o build/var_$NAME/cmx_pipeline.o $OBJS -ldl
echo built tools/ab/lib_$NAME.so
