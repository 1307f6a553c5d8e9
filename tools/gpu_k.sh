#!/bin/bash
mkdir -p gpurun_out/k
for v in 0 4 5; do
MPF_GEMM_VARIANT=$v python bench.py --random-start-leg 0 --no-cpu --steps 2 --warmup 1 --bootstrap-replicates 0 > gpurun_out/k/g$v.json 2>gpurun_out/k/g$v.err
python - $v <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/k/g{sys.argv[1]}.json"))
u = d["ufboot_online"]
print(sys.argv[1], u["roofline"]["frac"], u["roofline"]["kernel_ms_total"], u["seconds_each_pass"])
PY
done
MPF_GEMM_VARIANT=4 MPF_GEMM_SMALL=0 timeout 900 python -m pytest tests/test_gpu_ufboot.py -x -q 2>&1 | tail -2
