#!/usr/bin/env python3
"""Turn pmc_summary JSONs (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes and an SQ pass) into
profiles/traffic.json: per bench stage, HBM-side bytes per launch and VALU wave-instructions per launch.
Corrections (MI355X_MICROARCH.md, HBM section; re-checked with tools/calib_copy on this pool: a
512 MiB read reports FETCH_SIZE = 262146 KiB for 4-byte and for 16-byte loads):
read bytes = FETCH_SIZE * 1024 * 2, write bytes = WRITE_SIZE * 1024.
Every mode's entry is stamped with `batch` (frames per launch the passes were taken at) and `csrc_sha256`
(orbfe.source_hash() of the kernels measured): bench.py scales the counts by its own batch and DROPS them when
the sources have changed since ("if the kernels change and the PMC file is not refreshed, the
number silently lies").
usage: make_traffic_json.py <pmc_traffic.json> <pmc_sq.json> <mode> <out.json> [batch]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jetracer-orbslam2_amd"))
import orbfe  # noqa: E402  (source_hash only; nothing touches the GPU here)

src, sq, mode, out = sys.argv[1:5]
batch = int(sys.argv[5]) if len(sys.argv) > 5 else 256
d = json.load(open(src))
q = json.load(open(sq))

STAGES = {
    "pyramid": ["pyramid_fused_kernel", "blur_batch_kernel", "halfsample_batch_kernel"],
    "detect": ["detect_tile_kernel"],
    "describe": ["select_kernel", "describe_tile_kernel", "describe_kernel"],
    "align": ["align_fill_kernel", "align_splat_kernel", "align_unmax_kernel"],
    "match": ["match_gather_kernel", "match_batch_256_kernel", "match_batch_ref_kernel", "match_expand_kernel",
              "match_mfma_kernel", "match_tile_kernel", "match_bucket_kernel", "match_window_kernel"],
}


def per_launch(table, prefix, fn):
    tot = 0.0
    for k, v in table.items():
        if prefix in k:
            n = 7 if "halfsample" in k else 1  # 7 launches per step (levels 1..7); avg is per launch
            tot += n * fn(v)
    return tot


def bytes_of(v):
    return v.get("FETCH_SIZE", {}).get("avg", 0.0) * 1024 * 2 + v.get("WRITE_SIZE", {}).get("avg", 0.0) * 1024


stages = {s: sum(per_launch(d, p, bytes_of) for p in ks) for s, ks in STAGES.items()}
valu = {s: sum(per_launch(q, p, lambda v: v.get("SQ_INSTS_VALU", {}).get("avg", 0.0)) for p in ks) for s, ks in STAGES.items()}
try:
    allj = json.load(open(out))
except Exception:
    allj = {}
# fraction of SIMD cycles with a VALU instruction executing: SQ_ACTIVE_INST_VALU counts quad-cycles summed over
# the 1024 SIMDs, GRBM_GUI_ACTIVE the cycles of the 8 XCDs
act = {s: sum(per_launch(q, p, lambda v: v.get("SQ_ACTIVE_INST_VALU", {}).get("avg", 0.0)) for p in ks) for s, ks in STAGES.items()}
cyc = {s: sum(per_launch(q, p, lambda v: v.get("GRBM_GUI_ACTIVE", {}).get("avg", 0.0)) for p in ks) / 8.0 for s, ks in STAGES.items()}
stages["valu_wave_instructions"] = valu
# (an ISSUE COUNT, not a busy fraction: the counter charges every vector instruction a quad-cycle, full-rate instructions
# retire in 2.76 cycles here, so detect reads 1.06)
stages["valu_issue_quad_cycles_per_simd_cycle"] = {s: (act[s] * 4.0 / 1024.0 / cyc[s] if cyc[s] else None) for s in STAGES}
for name, counter in (("lds_bank_conflict_cycles", "SQ_LDS_BANK_CONFLICT"), ("salu_wave_instructions", "SQ_INSTS_SALU"),
                      ("lds_wave_instructions", "SQ_INSTS_LDS"), ("mfma_wave_instructions", "SQ_INSTS_MFMA")):
    stages[name] = {s: sum(per_launch(q, p, lambda v: v.get(counter, {}).get("avg", 0.0)) for p in ks) for s, ks in STAGES.items()}
stages["gpu_cycles"] = cyc
stages["batch"] = batch
stages["csrc_sha256"] = orbfe.source_hash()
allj[mode] = stages
allj["_note"] = ("per stage launch (`batch` frames): HBM-side bytes from rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE "
                 "passes (reads doubled per the gfx950 correction), VALU wave-instructions (SQ_INSTS_VALU) and the vector-instruction "
                 "issue quad-cycles per SIMD-cycle (SQ_ACTIVE_INST_VALU x 4 / 1024 / cycles; not a fraction) from SQ passes")
json.dump(allj, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(stages))
