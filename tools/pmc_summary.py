#!/usr/bin/env python3
"""Summarise the rocprofv3 counter CSVs written by tools/profile.sh into profiles/<tag>_pmc.json and profiles/traffic.json.

HBM traffic per launch follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly half the bytes of a wide (16 B/lane) coalesced read, so traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for d in sorted(glob.glob(os.path.join(root, "gpurun_out", f"pmc_{tag}_*"))):
    name = os.path.basename(d)[len(f"pmc_{tag}_"):]
    ctr = collections.defaultdict(list)
    kern = None
    for f in glob.glob(os.path.join(d, "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if ("ode_persistent_kernel" in r["Kernel_Name"]) if name == "persistent" else ("gemm" in r["Kernel_Name"]):
                kern = r["Kernel_Name"]
                ctr[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not ctr:
        continue
    avg = {k: sum(v[1:]) / max(1, len(v) - 1) for k, v in ctr.items()}   # drop the first (cold) launch
    s = {"kernel": kern, "launches_averaged": len(next(iter(ctr.values()))) - 1, "counters": {k: round(v, 1) for k, v in avg.items()}}
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
        s["hbm_bytes_per_launch"] = int((2 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024)
        s["fetch_bytes_corrected"] = int(2 * avg["FETCH_SIZE"] * 1024)
        s["write_bytes"] = int(avg["WRITE_SIZE"] * 1024)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "GRBM_GUI_ACTIVE" in avg:
        cyc = avg["GRBM_GUI_ACTIVE"] / 8.0                    # the counter sums the 8 XCDs
        s["kernel_cycles"] = round(cyc)
        s["mfma_util"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), 4)   # 256 CUs x 4 SIMDs
    if "TCC_HIT_sum" in avg:
        s["l2_hit_rate"] = round(avg["TCC_HIT_sum"] / max(1.0, avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"]), 4)
    out[name] = s
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
# what bench.py quotes as roofline.traffic: the dominant kernel's HBM bytes per launch, with the shape it was measured on
if "persistent" in out and "hbm_bytes_per_launch" in out["persistent"]:
    json.dump({"hbm_bytes_per_launch": out["persistent"]["hbm_bytes_per_launch"], "shape": ["persistent", 4096, 6, 512],
               "kernel": "ode_persistent_kernel<false>", "source": f"profiles/{tag}_pmc.json"},
              open(os.path.join(root, "profiles", "traffic.json"), "w"))
elif "fc1" in out and "hbm_bytes_per_launch" in out["fc1"]:
    json.dump({"hbm_bytes_per_launch": out["fc1"]["hbm_bytes_per_launch"], "shape": [4096, 2048, 512],
               "kernel": out["fc1"]["kernel"], "source": f"profiles/{tag}_pmc.json"},
              open(os.path.join(root, "profiles", "traffic.json"), "w"))
print(json.dumps(out, indent=1))
