#!/bin/bash
set -u
mkdir -p gpurun_out/c
A="--steps 20 --warmup 5 --no-cpu --bootstrap-replicates 0 --ufboot-samples 0 --random-start-leg 0 --weighted-leg 0 --start-trees 0 --climb-engines 0"
for o in "$@"; do
  python bench.py $A $o | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$o', 'step', round(d['ms_per_step'],4), 'scan', round(r['kernel_ms_per_launch'],4), 'plan', round(r['plan_kernel_ms_per_launch'],4), 'frac', round(r['frac'],3), 'views', round(d['views']['kernel_ms_per_step'],4), d['host_ms_per_step'])"
done
