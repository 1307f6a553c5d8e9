#!/bin/bash
set -u
mkdir -p gpurun_out/b
export TMPDIR=/tmp
A="--bootstrap-replicates 0 --ufboot-samples 0"
bash tools/pmc_quick.sh "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" $A > gpurun_out/b/pmc1.txt 2>&1
bash tools/pmc_quick.sh "SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" $A > gpurun_out/b/pmc2.txt 2>&1
bash tools/pmc_quick.sh "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" $A > gpurun_out/b/pmc3.txt 2>&1
bash tools/pmc_quick.sh "FETCH_SIZE WRITE_SIZE" $A > gpurun_out/b/pmc4.txt 2>&1
cat gpurun_out/b/pmc*.txt
python bench.py --steps 20 --warmup 5 --no-cpu $A | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['roofline']['plan_kernel_ms_per_launch'], d['views'], d['host_ms_per_step'])"
