#!/bin/bash
# k_bitgemm with parts of its loop removed (results wrong, timing only): which resource holds the kernel at ~0.48 of the MFMA peak
# the knock-out variants exist only in an experiments build of the library (results are wrong on purpose)
# (a library of its own beside the production one, selected through MPF_LIB_PATH: the shipped libmpfitch.so is never replaced)
EXP=$PWD/mpboot_amd/libmpfitch_exp.so
make -s -j8 -C mpboot_amd/csrc EXPERIMENTS=1 OUT=$EXP OBJDIR=$PWD/mpboot_amd/csrc/_obj_exp
export MPF_LIB_PATH=$EXP
mkdir -p gpurun_out/k
for e in 0 6 1 2 3 4; do
MPF_GEMM_EXPERIMENT=$e python bench.py --random-start-leg 0 --no-cpu --steps 2 --warmup 1 --bootstrap-replicates 0 > gpurun_out/k/x$e.json 2>gpurun_out/k/x$e.err
python - $e <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/k/x{sys.argv[1]}.json"))
u = d["ufboot_online"]
print("experiment", sys.argv[1], "frac", round(u["roofline"]["frac"], 4), "product kernels ms", round(u["roofline"]["kernel_ms_total"], 3))
PY
done
