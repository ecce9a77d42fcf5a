#!/usr/bin/env python3
"""profiles/traffic.json entry of bench.py --mode align from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over that bench:
HBM-side bytes per orbfe_align_depth_batch CALL = sum over its kernels of (average bytes per launch x launches per call)
(reads doubled: the gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md).  Stamped with the source hash like the other modes.
usage: make_align_traffic.py <pmc_summary.json> <calls profiled> <frames per call> <out traffic.json>"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jetracer-orbslam2_amd"))
import orbfe  # noqa: E402

src, calls, frames, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
d = json.load(open(src))
total, kernels = 0.0, {}
for k, v in d.items():
    if "align_" not in k:
        continue
    f, w = v.get("FETCH_SIZE", {}), v.get("WRITE_SIZE", {})
    n = max(f.get("n", 0), w.get("n", 0))
    b = (f.get("avg", 0.0) * 2048 + w.get("avg", 0.0) * 1024) * n / calls
    kernels[k.split("orbfe::")[-1][:40]] = {"launches_per_call": n / calls, "bytes_per_call": b}
    total += b
try:
    allj = json.load(open(out))
except Exception:
    allj = {}
allj["align"] = {"align": total, "batch": frames, "kernels": kernels, "csrc_sha256": orbfe.source_hash()}
json.dump(allj, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(allj["align"]))
