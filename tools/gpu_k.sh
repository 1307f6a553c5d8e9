#!/bin/bash
mkdir -p gpurun_out/k
timeout 2400 python -m pytest tests/test_gpu_ufboot.py tests/test_gpu_stateful.py tests/test_gpu_dropin.py -x -q > gpurun_out/k/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 30 gpurun_out/k/pytest.log
for off in 5000 6000; do MPF_FUZZ_OFFSET=$off timeout 600 python -m pytest tests/test_gpu_stateful.py -x -q 2>&1 | tail -n 2; done
