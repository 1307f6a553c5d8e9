#!/bin/bash
mkdir -p gpurun_out/k
bash tools/profile_gpu.sh r2 > gpurun_out/k/profile.log 2>&1; tail -n 2 gpurun_out/k/profile.log
python bench.py > gpurun_out/k/bench_default.json 2> gpurun_out/k/bench_default.err; cut -c1-330 gpurun_out/k/bench_default.json
