#!/bin/bash
mkdir -p gpurun_out/h
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/h/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 4 gpurun_out/h/pytest.log
python bench.py > gpurun_out/h/bench.json 2> gpurun_out/h/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/h/bench.json'))
print(d['ms_per_step'], d['ms_per_step_new_topology'], d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_launch'])
print(d['bootstrap_wall_clock']['seconds'], d['bootstrap_wall_clock']['refinement_s'], d['ufboot_online']['seconds_each_pass'])
print({k: (v if not isinstance(v, dict) else {a: b for a, b in v.items() if a != 'cpu_baseline'}) for k, v in d['random_start'].items()})
PY
