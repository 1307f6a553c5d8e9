#!/bin/bash
mkdir -p gpurun_out/k
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_stateful.py tests/test_gpu_parity.py -x -q > gpurun_out/k/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/k/pytest.log
for i in 1 2; do python bench.py --random-start-leg 0 --bootstrap-replicates 0 --ufboot-samples 0 --no-cpu > gpurun_out/k/b$i.json 2>gpurun_out/k/b$i.err
python - gpurun_out/k/b$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["ms_per_step"], d["ms_per_step_new_topology"], d["value"], d["host_ms_per_step"])
PY
done
