#!/bin/bash
for q in 4 8 16 24; do
  echo "GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q python tools/concurrent_climbs.py --engines 4,8,16 2>&1 | tail -3 | cut -c1-90
done
for q in 4 16; do
GPU_MAX_HW_QUEUES=$q python bench.py --random-start-leg 0 --no-cpu --steps 2 --warmup 1 --engines-per-gpu 8 > /tmp/r$q.json 2>/dev/null
python - $q <<'PY'
import json, sys
d = json.load(open(f"/tmp/r{sys.argv[1]}.json")); b = d["bootstrap_wall_clock"]
print("queues", sys.argv[1], "refinement", b["refinement_s"], b["refinement_s_plan_cache_off"], "engines", b["engines_per_gpu"])
PY
done
