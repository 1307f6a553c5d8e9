#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_edges.py -x -q 2>&1 | tail -4
bash tools/gpu_c.sh "--opt plan_cache=1" "--opt plan_cache=0"
