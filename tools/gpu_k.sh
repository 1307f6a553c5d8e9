#!/bin/bash
mkdir -p gpurun_out/k
show () { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["ms_per_step"], d["ms_per_step_new_topology"], d["host_ms_per_step"])
PY
}
for i in 1 2; do python bench.py --random-start-leg 0 --bootstrap-replicates 0 --ufboot-samples 0 --no-cpu > gpurun_out/k/b$i.json 2>gpurun_out/k/b$i.err; show gpurun_out/k/b$i.json; done
MPF_HOST_POLL=0 python bench.py --random-start-leg 0 --bootstrap-replicates 0 --ufboot-samples 0 --no-cpu > gpurun_out/k/b3.json 2>gpurun_out/k/b3.err; show gpurun_out/k/b3.json
