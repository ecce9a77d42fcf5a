#!/usr/bin/env python3
"""Turn a pmc_summary JSON (FETCH_SIZE / WRITE_SIZE per kernel, separate rocprofv3 --pmc
passes) into profiles/traffic.json: HBM-side bytes per launch of each bench stage.
Corrections (MI355X_MICROARCH.md, HBM section; re-checked with tools/calib_copy on this
pool: a 512 MiB read reports FETCH_SIZE = 262146 KiB for 4-byte and for 16-byte loads):
read bytes = FETCH_SIZE * 1024 * 2, write bytes = WRITE_SIZE * 1024.
usage: make_traffic_json.py <pmc_summary.json> <mode> <out.json>"""
import json
import sys

src, mode, out = sys.argv[1:4]
d = json.load(open(src))


def b(prefix):
    tot = 0.0
    for k, v in d.items():
        if prefix in k:
            n = 1
            r = v.get("FETCH_SIZE", {}).get("avg", 0.0) * 1024 * 2
            w = v.get("WRITE_SIZE", {}).get("avg", 0.0) * 1024
            if "halfsample" in k:
                n = 7  # 7 launches per step (levels 1..7); avg is per launch
            tot += n * (r + w)
    return tot


stages = {
    "pyramid": b("pyramid_fused_kernel") + b("blur_batch_kernel") + b("halfsample_batch_kernel"),
    "detect": b("detect_tile_kernel"),
    "describe": b("select_kernel") + b("describe_kernel"),
    "match": b("match_gather_kernel") + b("match_batch_256_kernel") + b("match_batch_ref_kernel")
             + b("match_expand_kernel") + b("match_mfma_kernel"),
}
try:
    allj = json.load(open(out))
except Exception:
    allj = {}
allj[mode] = stages
allj["_note"] = ("HBM-side bytes per stage launch (256 frames), from rocprofv3 --pmc FETCH_SIZE and "
                 "--pmc WRITE_SIZE passes; reads doubled per the gfx950 correction")
json.dump(allj, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(stages))
