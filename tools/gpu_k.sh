#!/bin/bash
mkdir -p gpurun_out/k
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_stateful.py tests/test_gpu_edges.py -x -q > gpurun_out/k/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/k/pytest.log
bash tools/profile_gpu.sh r2 > gpurun_out/k/profile.log 2>&1; tail -n 3 gpurun_out/k/profile.log
python bench.py > gpurun_out/k/bench_default.json 2> gpurun_out/k/bench_default.err; cut -c1-400 gpurun_out/k/bench_default.json
