#!/bin/bash
# sweep the front-end gather's events-per-workgroup (rebuilds the library per configuration); run on the GPU box
cd $(dirname $0)/..
for cfg in "1536 768" "1024 2048" "512 4096" "2048 1024" "768 2048"; do
  set -- $cfg
  sed -i "s/^constexpr int kFeGatherPerBlock = .*/constexpr int kFeGatherPerBlock = $1, kFeGatherCap = $2;/" cmax_slam_amd/csrc/cmx_kernels.hip
  make -C cmax_slam_amd/csrc -s 2>&1 | grep -E "error" | head -2
  python tools/kernel_times.py 2>&1 | grep "kernel us" | sed "s/^/per_block $1 cap $2: /" | cut -c1-150
done
