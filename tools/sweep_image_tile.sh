#!/bin/bash
# sweep tile shapes of image_adjoint_kernel (rebuilds the library per configuration); run on the GPU box
cd $(dirname $0)/..
for cfg in "64 16 1024" "64 8 512" "32 16 512" "64 32 1024" "128 8 1024" "32 32 1024" "64 16 512" "32 8 256"; do
  set -- $cfg
  sed -i "s/^constexpr int kAdjTX = .*/constexpr int kAdjTX = $1, kAdjTY = $2, kAdjThreads = $3;/" cmax_slam_amd/csrc/cmx_kernels.hip
  make -C cmax_slam_amd/csrc -s 2>&1 | grep -E "error" | head -2
  python tools/kernel_times.py 2>&1 | grep "kernel us" | sed "s/^/tile $1x$2 threads $3: /" | cut -c1-150
done
