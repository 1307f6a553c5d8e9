#!/bin/bash
mkdir -p gpurun_out/e
timeout 600 python -m pytest tests/test_gpu_dropin.py -x -q 2>&1 | tail -30 > gpurun_out/e/dropin.log
python bench.py > gpurun_out/e/bench.json 2> gpurun_out/e/bench.err; echo bench rc=$?
tail -25 gpurun_out/e/dropin.log; tail -3 gpurun_out/e/bench.err
