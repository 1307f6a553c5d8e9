#!/bin/bash
# What bounds k_scan_prog: the kernel, its loads-only and arithmetic-only variants, the same on the row-major
# store (four row loads per vector) and with every load confined to two hot vectors, the walking kernel, the VALU issue rates of
# the instructions it is made of and the load rate of a CU by access shape.
# the knock-out variants exist only in an experiments build of the library (results are wrong on purpose)
# (a library of its own beside the production one, selected through MPF_LIB_PATH: the shipped libmpfitch.so is never replaced)
EXP=$PWD/mpboot_amd/libmpfitch_exp.so
make -s -j8 -C mpboot_amd/csrc EXPERIMENTS=1 OUT=$EXP OBJDIR=$PWD/mpboot_amd/csrc/_obj_exp
export MPF_LIB_PATH=$EXP
mkdir -p gpurun_out/bounds
for u in valu_rate l1_rate; do hipcc --offload-arch=gfx950 -O3 -o tools/ubench/$u tools/ubench/$u.hip; done
{
echo "== bench.py --steps 20 --warmup 5 (C3 sweep; scan = HIP-event time of the scan kernel per launch, ms) =="
echo "-- the kernel as shipped: child vectors from the word-major copy, one buffer_load_dwordx4 per vector and lane"
bash tools/bench_oneline.sh "--opt scan_prog=1"
echo "-- MPF_PROG_EXPERIMENT=2: loads only (one AND per loaded register instead of the Fitch arithmetic; results are garbage on purpose)"
MPF_PROG_EXPERIMENT=2 bash tools/bench_oneline.sh "--opt scan_prog=1"
echo "-- MPF_PROG_EXPERIMENT=1: arithmetic + control only (no vector loads in the loop)"
MPF_PROG_EXPERIMENT=1 bash tools/bench_oneline.sh "--opt scan_prog=1"
echo "-- scan_shadow=0: child vectors from the row-major store, four buffer_load_dword per vector and lane (round 2's kernel)"
bash tools/bench_oneline.sh "--opt scan_prog=1 --opt scan_shadow=0"
echo "-- ... loads only"
MPF_PROG_EXPERIMENT=2 bash tools/bench_oneline.sh "--opt scan_prog=1 --opt scan_shadow=0"
echo "-- ... host-planned (dev_plan=0) with MPF_PROG_CID_MASK=1: every child vector one of TWO hot ones (vector-L1 hits)"
MPF_PROG_CID_MASK=1 bash tools/bench_oneline.sh "--opt scan_prog=1 --opt scan_shadow=0 --opt dev_plan=0"
echo "-- ... the same, loads only"
MPF_PROG_EXPERIMENT=2 MPF_PROG_CID_MASK=1 bash tools/bench_oneline.sh "--opt scan_prog=1 --opt scan_shadow=0 --opt dev_plan=0"
echo "-- scan_prog=0: the device-walked kernel of round 1 (with long neighbourhoods cut, and as it was)"
bash tools/bench_oneline.sh "--opt scan_prog=0" "--opt scan_prog=0 --opt split_cands=100000"
echo
echo "== tools/ubench/valu_rate: ns per wave-instruction per SIMD with 1/2/4/8 waves per SIMD =="
tools/ubench/valu_rate
echo
echo "== tools/ubench/l1_rate: bytes a CU loads per second from a hot set of 1-KB vectors, by access shape =="
tools/ubench/l1_rate
} > gpurun_out/bounds/scan_bounds.txt 2>&1
python tools/wave_timeline.py > gpurun_out/bounds/wave_timeline.txt 2>&1
cat gpurun_out/bounds/scan_bounds.txt
