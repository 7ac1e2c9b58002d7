#!/bin/bash
# Round 5: front-end launch shape at 1M events -- threads per workgroup x events per workgroup for the gather
# (tail finalize on), threads per chunk workgroup x chunk size for the splat.  Builds the variants HERE (hipcc cross-compiles);
# run them on the GPU box in one call:  tools/ab_builds.sh "fe reps=300" 3
set -e
cd "$(dirname "$0")/.."
rm -f tools/ab/lib_*.so
tools/build_variant.sh a_base ""
for cfg in "256 2048" "256 4096" "512 1024" "512 2048" "512 4096" "1024 2048" "1024 4096"; do
  set -- $cfg
  tools/build_variant.sh g_nt$1_per$2 "-DCMX_FE_GATHER_NT=$1 -DCMX_FE_GATHER_PER_BLOCK=$2"
done
for cfg in "256 384" "256 256" "512 512" "512 384" "512 256" "1024 512" "1024 256" "1024 192"; do
  set -- $cfg
  tools/build_variant.sh s_nt$1_div$2 "-DCMX_FE_SPLAT_NT=$1 -DCMX_FE_CHUNK_DIV=$2"
done
ls -la tools/ab/
