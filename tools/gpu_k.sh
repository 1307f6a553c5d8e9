#!/bin/bash
mkdir -p gpurun_out/k
MPF_EXTREMES_FIRST=3 timeout 300 python -u tools/extremes_probe.py > gpurun_out/k/extremes.log 2>&1; echo "rc=$?"; tail -n 25 gpurun_out/k/extremes.log | cut -c1-300
