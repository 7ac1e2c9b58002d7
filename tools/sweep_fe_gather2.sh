#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "1024 2048 2" "1536 2048 2" "2048 1024 2" "1024 2048 2" "3072 1024 2"; do
  set -- $cfg
  sed -i "s/^constexpr int kFeGatherPerBlock = .*/constexpr int kFeGatherPerBlock = $1, kFeGatherCap = $2;/" cmax_slam_amd/csrc/cmx_kernels.hip
  sed -i "s/^  constexpr int U = 2;  \/\/ events in flight per thread (swept/  constexpr int U = $3;  \/\/ events in flight per thread (swept/" cmax_slam_amd/csrc/cmx_kernels.hip
  make -C cmax_slam_amd/csrc -s 2>&1 | grep -E "error" | head -2
  for r in 1 2 3; do python bench.py --steps 300 --no-cpu-baseline --solves 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('per_block $1 U $3:', round(d['ms_per_step']*1e3,2), {k:round(x*1e3,2) for k,x in d['kernel_ms'].items()})"; done
  sed -i "s/^  constexpr int U = $3;  \/\/ events in flight per thread (swept/  constexpr int U = 2;  \/\/ events in flight per thread (swept/" cmax_slam_amd/csrc/cmx_kernels.hip
done
