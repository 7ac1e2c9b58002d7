#!/bin/bash
# Runs on the GPU box (via gpurun): PMC passes over tools/ab_eval.py be tile=1 (tile-ordered vs time-ordered back-end gather).
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_tile
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/ab_eval.py be tile=1 reps=20"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp -d $OUT/p$i -o pmc -- $CMD > $OUT/p$i.log 2>&1
done
cd $REPO
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -E "be_gather|be_splat_lds" $OUT/summary.txt | cut -c1-150
find $OUT -name "*.db" -size +20M -delete
