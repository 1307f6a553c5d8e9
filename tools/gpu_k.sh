#!/bin/bash
mkdir -p gpurun_out/k
for n in 1 2 4 6 8; do
python bench.py --random-start-leg 0 --no-cpu --steps 2 --warmup 1 --engines-per-gpu $n > gpurun_out/k/r$n.json 2>gpurun_out/k/r$n.err
python - $n <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/k/r{sys.argv[1]}.json"))
b = d["bootstrap_wall_clock"]
print(sys.argv[1], b["refinement_s"], b["online_phase_s"], b["seconds"])
PY
done
