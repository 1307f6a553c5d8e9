#!/bin/bash
set -u
mkdir -p gpurun_out/full
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/full/pytest.log 2>&1
echo "pytest rc=$?" > gpurun_out/full/rc.txt
timeout 900 python bench.py > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err
echo "bench rc=$?" >> gpurun_out/full/rc.txt
cat gpurun_out/full/rc.txt; tail -n 5 gpurun_out/full/pytest.log; cut -c1-600 gpurun_out/full/bench.json; tail -n 3 gpurun_out/full/bench.err
