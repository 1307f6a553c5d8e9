#!/bin/bash
# What bounds k_scan_prog: the kernel, its loads-only and arithmetic-only variants, the same with every
# load confined to 64 hot vectors, the walking kernel, and the VALU issue rates of the instructions it is made of.
# the knock-out variants exist only in an experiments build of the library (results are wrong on purpose)
make -s -C mpboot_amd/csrc clean && make -s -j8 -C mpboot_amd/csrc EXPERIMENTS=1
trap 'make -s -C mpboot_amd/csrc clean && make -s -j8 -C mpboot_amd/csrc' EXIT
mkdir -p gpurun_out/bounds
{
echo "== bench.py --steps 20 --warmup 5 (C3 sweep; scan = HIP-event time of the scan kernel per launch, ms) =="
bash tools/bench_oneline.sh "--opt scan_prog=1"
echo "-- MPF_PROG_EXPERIMENT=2: loads only (one AND per loaded register instead of the Fitch arithmetic; results are garbage on purpose)"
MPF_PROG_EXPERIMENT=2 bash tools/bench_oneline.sh "--opt scan_prog=1"
echo "-- MPF_PROG_EXPERIMENT=1: arithmetic + control only (no vector loads in the loop)"
MPF_PROG_EXPERIMENT=1 bash tools/bench_oneline.sh "--opt scan_prog=1"
echo "-- MPF_PROG_CID_MASK=63: the real kernel, every child vector taken from 64 hot ones"
MPF_PROG_CID_MASK=63 bash tools/bench_oneline.sh "--opt scan_prog=1"
echo "-- scan_prog=0: the device-walked kernel of round 1 (with long neighbourhoods cut, and as it was)"
bash tools/bench_oneline.sh "--opt scan_prog=0" "--opt scan_prog=0 --opt split_cands=100000"
echo
echo "== tools/ubench/valu_rate: ns per wave-instruction per SIMD with 1/2/4/8 waves per SIMD =="
tools/ubench/valu_rate
} > gpurun_out/bounds/scan_bounds.txt 2>&1
python tools/wave_timeline.py > gpurun_out/bounds/wave_timeline.txt 2>&1
python tools/climb_timing.py --workload C3 --start random > gpurun_out/bounds/climb_c3_random.txt 2>&1
python tools/climb_timing.py --workload C2 --start random --check > gpurun_out/bounds/climb_c2_random.txt 2>&1
python tools/hbm_copy_bw.py > gpurun_out/bounds/hbm_copy.txt 2>&1
cat gpurun_out/bounds/scan_bounds.txt
