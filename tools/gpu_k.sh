#!/bin/bash
mkdir -p gpurun_out/k
timeout 1500 python -m pytest tests/test_gpu_ufboot.py tests/test_gpu_stateful.py tests/test_gpu_configs.py tests/test_gpu_dropin.py -x -q > gpurun_out/k/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/k/pytest.log
MPF_UFB_PROFILE=1 python bench.py --no-cpu --steps 3 --warmup 1 --bootstrap-replicates 0 2>gpurun_out/k/e.err > gpurun_out/k/e.json
grep -E "ufboot\]" gpurun_out/k/e.err | cut -c1-200
python - <<'PY'
import json
d=json.load(open("gpurun_out/k/e.json")); print(d["ufboot_online"]["seconds_each_pass"], d["random_start"]["bb_flow"]["online_phase_s"])
PY
