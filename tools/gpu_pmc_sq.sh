#!/bin/bash
# Runs on the GPU box (via gpurun): instruction-mix / stall PMC passes of the bench command (one counter group per pass).
# Usage: tools/gpu_pmc_sq.sh <tag> [bench args...]
# (a counter group the hardware cannot collect in one pass makes rocprofv3 abort and then hang: every pass is under timeout)
set -u
TAG=${1:-sq}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 30 --warmup 5 --no-cpu-baseline --solves 0 $*"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $grp -d $OUT/p$i -o pmc -- $BENCH > $OUT/p$i.log 2>&1
done
cd $REPO
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -E "fe_gather|fe_splat_lds|image_adjoint|be_gather_kernel|be_splat_lds|be_pose" $OUT/summary.txt | cut -c1-140
find $OUT -name "*.db" -size +20M -delete
