#!/bin/bash
mkdir -p gpurun_out/k
timeout 600 python -m pytest tests/test_gpu_edges.py tests/test_gpu_ufboot.py -x -q -k "ladder or thousands or millions or heavy or third_plane or tile_edges" > gpurun_out/k/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 30 gpurun_out/k/pytest.log
