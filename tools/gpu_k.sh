#!/bin/bash
mkdir -p gpurun_out/k
run () { echo "== $*"; env "$@" python tools/ufboot_timing.py --workload C3 --samples 1000 --verify 0 --start random 2>&1 | grep -E "climb with online" | sed -e 's/.*climb with online UFBoot/online/' | cut -c1-200; }
run MPF_GEMM_WANT=128 MPF_GEMM_SMALL=1
run MPF_GEMM_WANT=64 MPF_GEMM_SMALL=1
run MPF_GEMM_WANT=64
run MPF_GEMM_WANT=192 MPF_GEMM_SMALL=1
