#!/usr/bin/env python3
"""profiles/rN/traffic.json from the PMC passes of tools/profile_gpu.sh: bytes that left the L2s per launch, per kernel family
(FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 -- the counter tallies 128-B requests at 64 B -- plus
WRITE_SIZE, both in KB), tied to a hash of the device sources so that bench.py quotes them only for the build they describe."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (source_hash)

out = sys.argv[1]
res = {"workload": "C3", "source_hash": bench.source_hash(), "fetch_correction": 2.0,
       "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_gpu.sh (main leg: python3 bench.py --steps 5 --warmup 1 "
                 "--no-cpu --bootstrap-replicates 0 --ufboot-samples 0 --random-start-leg 0)", "kernels": {}}
for leg in ("main", "ufboot"):
    path = os.path.join(out, leg, "summary.txt")
    if not os.path.exists(path):
        continue
    vals = {}
    for line in open(path):
        m = re.match(r"\s+(.*?)\s+((?:FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum|TCC_MISS_sum)=.*)$", line)
        if not m:
            continue
        name = m.group(1).strip()
        for tok in m.group(2).split():
            k, v = tok.split("=", 1)
            vals.setdefault(name, {})[k] = float(v.split("(")[0])
            vals[name]["n_" + k] = int(v.split("n=")[1].rstrip(")")) if "n=" in v else None
    for name, v in vals.items():
        if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        short = re.sub(r"^void\s+", "", name).replace("mpf::", "")
        key = short.split("<")[0].split("(")[0]
        if leg == "ufboot" and not key.startswith("k_bitgemm") and not key.startswith("k_ufb"):
            continue
        if key in res["kernels"]:
            continue
        res["kernels"][key] = {"leg": leg, "kernel": short[:80], "fetch_size_kb_raw": v["FETCH_SIZE"], "write_size_kb": v["WRITE_SIZE"],
                               "bytes_per_launch": int((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024),
                               "tcc_hit": v.get("TCC_HIT_sum"), "tcc_miss": v.get("TCC_MISS_sum"), "dispatches": v.get("n_FETCH_SIZE")}
print(json.dumps(res, indent=1))
