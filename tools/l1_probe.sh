#!/bin/bash
# Is k_scan_prog bound by the L2 -> L1 fill path or by the load path inside the CU?  The real kernel with every child vector taken
# from 64 / 8 / 2 hot vectors (the last two sets fit the 32 KB vector cache of a CU many times over).  Experiments build only.
make -s -C mpboot_amd/csrc clean && make -s -j8 -C mpboot_amd/csrc EXPERIMENTS=1
trap 'make -s -C mpboot_amd/csrc clean && make -s -j8 -C mpboot_amd/csrc' EXIT
mkdir -p gpurun_out/bounds
{
echo "== k_scan_prog, C3 sweep, host-planned (dev_plan 0) so that the plan kernel applies the mask; scan = HIP-event ms per launch =="
for m in 0xFFFFFFFF 63 7 1; do
  echo "-- MPF_PROG_CID_MASK=$m"
  MPF_PROG_CID_MASK=$m bash tools/bench_oneline.sh "--opt scan_prog=1 --opt dev_plan=0"
  echo "-- MPF_PROG_CID_MASK=$m, loads only (MPF_PROG_EXPERIMENT=2)"
  MPF_PROG_EXPERIMENT=2 MPF_PROG_CID_MASK=$m bash tools/bench_oneline.sh "--opt scan_prog=1 --opt dev_plan=0"
done
} > gpurun_out/bounds/l1_probe.txt 2>&1
cat gpurun_out/bounds/l1_probe.txt
