#!/bin/bash
# Runs on the GPU box (via gpurun): PMC passes over tools/ab_eval.py (one counter group per pass).
# Usage: tools/gpu_pmc_ab.sh <tag> [ab_eval args...]
set -u
TAG=${1:-ab}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/ab_eval.py $* reps=30"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $grp -d $OUT/p$i -o pmc -- $CMD > $OUT/p$i.log 2>&1
done
cd $REPO
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -E "fe_fused|fe_gather|fe_splat_lds|image_adjoint|image_moments" $OUT/summary.txt | cut -c1-150
find $OUT -name "*.db" -size +20M -delete
