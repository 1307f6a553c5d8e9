#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + separate PMC passes of the bench command.
# Usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --no-cpu $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python3 bench.py $ARGS > $OUT/bench_pmc_$N.json 2> $OUT/pmc_$N.err
done
python3 tools/profile_summary.py $OUT > $OUT/summary.txt 2>&1
# keep only the small artefacts (the raw per-dispatch CSVs exceed gpurun's 64 MiB return limit)
mkdir -p $OUT/keep
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/keep/kernel_stats.csv \;
rm -rf $OUT/trace $OUT/pmc_*
cat $OUT/summary.txt
