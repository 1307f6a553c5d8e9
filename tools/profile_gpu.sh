#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of the bench command.
#   main leg   : bench.py --steps 5 --warmup 1 with the bootstrap/UFBoot legs off, so that every k_scan_walk dispatch is
#                one of the timed sweep scans and the per-kernel averages are comparable with bench.py's own HIP-event timing
#   ufboot leg : bench.py --steps 1 --warmup 0 with only the online-UFBoot leg on (k_bitgemm and friends)
# Usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export MPF_BENCH_LIVE_TRAFFIC=0     # bench.py's own counter passes are child processes: not from a process the profiler has given the GPU
run_leg () {   # name, bench args, counter groups...
  local NAME=$1 ARGS=$2; shift 2
  local D=$OUT/$NAME
  mkdir -p $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 bench.py $ARGS > $D/bench_trace.json 2> $D/trace.err
  for C in "$@"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-60)
    rocprofv3 --pmc $C --output-format csv -d $D/pmc_$N -- python3 bench.py $ARGS > $D/bench_pmc_$N.json 2> $D/pmc_$N.err
  done
  python3 tools/profile_summary.py $D > $D/summary.txt 2>&1
  # keep only the small artefacts (the raw per-dispatch CSVs exceed gpurun's 64 MiB return limit)
  find $D/trace -name "*kernel_stats.csv" -exec cp {} $D/kernel_stats.csv \;
  rm -rf $D/trace $D/pmc_*
}
run_leg main "--legs none --steps 5 --warmup 1 --no-cpu --bootstrap-replicates 0 --ufboot-samples 0 --random-start-leg 0 --weighted-leg 0 --start-trees 0 --climb-engines 0 --tree-cache $OUT/tree $*" \
  FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
  "SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" GRBM_GUI_ACTIVE
run_leg ufboot "--legs none --steps 1 --warmup 0 --no-cpu --bootstrap-replicates 0 --random-start-leg 0 --weighted-leg 0 --start-trees 0 --climb-engines 0 --tree-cache $OUT/tree $*" \
  FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_BUSY_CYCLES" \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" GRBM_GUI_ACTIVE
python3 tools/make_traffic.py $OUT > $OUT/traffic.json
cat $OUT/main/summary.txt $OUT/ufboot/summary.txt
