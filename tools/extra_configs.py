#!/usr/bin/env python3
"""Single-GPU throughput of the batch path at the other shapes BASELINE.md lists (parity cases, not
the metric): C3 848x480 pairs, C4 1280x720 x 64, C5 3840x2160 with 12 levels and 8000 features.
Prints one line per shape: frames/s, keypoints/s, ms per batch (HIP events, inputs resident in HBM)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jetracer-orbslam2_amd"))
import orbfe  # noqa: E402
from orbfe import synth  # noqa: E402

SHAPES = [
    ("C3 848x480, 8 levels, 2000 features, batch 2 (one pair)", 848, 480, 2, dict(levels=8, cell=8, min_arc=9, max_features=2000)),
    ("C3 848x480, 8 levels, 2000 features, batch 256", 848, 480, 256, dict(levels=8, cell=8, min_arc=9, max_features=2000)),
    ("C4 1280x720, 8 levels, 2000 features, batch 64", 1280, 720, 64, dict(levels=8, cell=8, min_arc=9, max_features=2000)),
    ("C5 3840x2160, 12 levels, 8000 features, batch 8", 3840, 2160, 8, dict(levels=12, cell=16, min_arc=9, max_features=8000)),
]
s = torch.cuda.current_stream().cuda_stream
for name, w, h, B, cfg in SHAPES:
    ctx = orbfe.Context(w, h, max_batch=B, **cfg)
    nd = min(B, 4)
    base = synth.frames(w, h, nd, first_index=7, kind="rects", n_rects=800 * (w * h) // (640 * 480), min_size=6, max_size=32)
    frames = torch.from_numpy(base).cuda()[torch.arange(B, device="cuda") % nd].contiguous()
    rec = torch.zeros(B * ctx.cap * 52, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    idx = torch.zeros(max(B - 1, 1) * ctx.cap, dtype=torch.int32, device="cuda")
    dst = torch.zeros(max(B - 1, 1) * ctx.cap, dtype=torch.int32, device="cuda")

    def step():
        ctx.extract(frames.data_ptr(), w, w * h, B, rec.data_ptr(), cnt.data_ptr(), None, s)
        ctx.match_batch(rec.data_ptr(), cnt.data_ptr(), B, 1, -1, 256, idx.data_ptr(), dst.data_ptr(), s)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    kp = int(cnt.sum().item())
    print("%-58s %9.0f frames/s  %.3g keypoints/s  %.3f ms per batch  (%d keypoints/frame)"
          % (name, B / ms * 1e3, kp / ms * 1e3, ms, kp // B))
    ctx.close()
    del frames, rec, cnt, idx, dst
    torch.cuda.empty_cache()
