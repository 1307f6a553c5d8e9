#!/bin/bash
mkdir -p gpurun_out/i
timeout 1200 python -m pytest tests/test_gpu_ufboot.py tests/test_gpu_dropin.py tests/test_abi.py -x -q > gpurun_out/i/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 15 gpurun_out/i/pytest.log
