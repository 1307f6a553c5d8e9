#!/bin/bash
mkdir -p gpurun_out/k
for off in 7000 8000 9000; do MPF_FUZZ_OFFSET=$off timeout 900 python -m pytest tests/test_gpu_stateful.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -n 2; done
for t in 32 8; do MPF_VIEWS_TILE=$t timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_edges.py -x -q 2>&1 | tail -n 1; done
