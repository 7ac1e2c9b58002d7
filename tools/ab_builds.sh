#!/bin/bash
# Same-box A/B of several builds of libcmaxhip.so (tools/ab/lib_*.so, built here and carried along by gpurun):
#   tools/ab_builds.sh "<ab_eval args>" [rounds]     e.g. tools/ab_builds.sh "be reps=300" 3
# Box-to-box differences are 1-2 us on a 40 us kernel: variants are only comparable inside one call.
ARGS=${1:-"be reps=300"}; ROUNDS=${2:-3}
for r in $(seq $ROUNDS); do
  for so in tools/ab/lib_*.so; do
    echo -n "$(basename $so .so)  "
    CMAX_HIP_SO=$PWD/$so python tools/ab_eval.py $ARGS 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/ c=.*fdf kernels/ fdf kernels/' | cut -c1-220
  done
done
