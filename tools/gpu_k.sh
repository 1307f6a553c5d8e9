#!/bin/bash
mkdir -p gpurun_out/k
python tools/ufboot_timing.py --workload C3 --samples 1000 --verify 0 --start random --storetrees 2>&1 | cut -c1-400 | tail -5
python tools/ufboot_timing.py --workload C2 --samples 1000 --verify 2 --start random --storetrees --check 2>&1 | cut -c1-400 | tail -5
bash tools/profile_gpu.sh r2 > gpurun_out/k/profile.log 2>&1; tail -n 5 gpurun_out/k/profile.log
python bench.py > gpurun_out/k/bench_default.json 2> gpurun_out/k/bench_default.err; cut -c1-600 gpurun_out/k/bench_default.json
