#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes of the default bench command.
# Usage: tools/gpu_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
# what is being measured: source hash of cmax_slam_amd/csrc + hash of the library, read back by tools/pmc_to_json.py
python -c "import sys; sys.path.insert(0, '$REPO'); import bench, hashlib; print(bench.csrc_hash()); print(hashlib.sha256(open('$REPO/cmax_slam_amd/libcmaxhip.so','rb').read()).hexdigest())" > $OUT/stamp.txt
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-per-packet $*"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc TCC_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_tcc -o pmc -- $BENCH > $OUT/pmc_tcc.log 2>&1
# instruction issue (round 5: the back-end kernels are priced against the VALU pipe that binds them, bench.py roofline.bound "valu_fp64")
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD -d $OUT/pmc_sq_insts -o pmc -- $BENCH > $OUT/pmc_sq_insts.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d $OUT/pmc_sq_active -o pmc -- $BENCH > $OUT/pmc_sq_active.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/pmc_sq_waves -o pmc -- $BENCH > $OUT/pmc_sq_waves.log 2>&1
# instruction MIX (round 6: the VALU bound is priced per class, tools/microbench/valu_rates.hip: everything 4 clocks but TRANS_F32 8 / TRANS_F64 16)
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 -d $OUT/pmc_mix_f64 -o pmc -- $BENCH > $OUT/pmc_mix_f64.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 -d $OUT/pmc_mix_f32 -o pmc -- $BENCH > $OUT/pmc_mix_f32.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d $OUT/pmc_mix_int -o pmc -- $BENCH > $OUT/pmc_mix_int.log 2>&1
# FETCH_SIZE calibration on known byte counts in the kernels' own access widths (tools/microbench/fetch_calib.hip)
if [ -x $REPO/tools/microbench/fetch_calib ]; then
  timeout 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_calib -o pmc -- $REPO/tools/microbench/fetch_calib > $OUT/pmc_calib.log 2>&1
fi
cd $REPO
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep only the small files
find $OUT -name "*.db" -size +20M -delete
