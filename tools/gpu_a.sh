#!/bin/bash
# round-2 first GPU pass: the new planned-program scan against the pinned tests, then the bench
set -u
mkdir -p gpurun_out/a
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "scan_candidates or hill_climb or long_climb or native" > gpurun_out/a/parity.log 2>&1
echo "parity rc=$?" > gpurun_out/a/rc.txt
timeout 1500 python -m pytest tests/test_gpu_configs.py -x -q > gpurun_out/a/configs.log 2>&1
echo "configs rc=$?" >> gpurun_out/a/rc.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --bootstrap-replicates 0 --ufboot-samples 0 > gpurun_out/a/bench_prog.json 2> gpurun_out/a/bench_prog.err
echo "bench rc=$?" >> gpurun_out/a/rc.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --bootstrap-replicates 0 --ufboot-samples 0 --opt scan_prog=0 > gpurun_out/a/bench_walk.json 2> gpurun_out/a/bench_walk.err
echo "bench walk rc=$?" >> gpurun_out/a/rc.txt
cat gpurun_out/a/rc.txt
tail -n 5 gpurun_out/a/parity.log; tail -n 5 gpurun_out/a/configs.log
cat gpurun_out/a/bench_prog.json gpurun_out/a/bench_walk.json | cut -c1-1500
