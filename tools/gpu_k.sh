#!/bin/bash
mkdir -p gpurun_out/k
timeout 1500 python -m pytest tests/test_gpu_ufboot.py tests/test_gpu_stateful.py tests/test_gpu_dropin.py -x -q > gpurun_out/k/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/k/pytest.log
MPF_UFB_PROFILE=1 python tools/ufboot_timing.py --workload C3 --samples 1000 --verify 0 --start random 2>&1 | cut -c1-260 | tail -8
