#!/usr/bin/env python3
"""Determinism soak: run extraction + matching on the same resident batch many times and require every
output byte to repeat (a race in a queue, an atomic or a barrier shows up as a changed checksum).
usage: soak.py [iters]   -- C2, C3 (windowed), reference and C5 shapes, small batches"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jetracer-orbslam2_amd"))
import orbfe  # noqa: E402
from orbfe import synth  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
CASES = [
    ("c2", 640, 480, 32, dict(levels=8, cell=8, min_arc=9, max_features=2000), dict(mode=1, window=-1, md=64, stride=1)),
    ("c3", 848, 480, 32, dict(levels=8, cell=8, min_arc=9, max_features=2000), dict(mode=1, window=16, md=80, stride=2)),
    ("ref", 640, 480, 32, dict(), dict(mode=0, window=32, md=8, stride=1)),
    ("c5", 3840, 2160, 2, dict(levels=12, cell=16, min_arc=9, max_features=8000), dict(mode=1, window=-1, md=64, stride=1)),
]
s = torch.cuda.current_stream().cuda_stream
bad = 0
for name, w, h, B, cfg, mm in CASES:
    ctx = orbfe.Context(w, h, max_batch=B, **cfg)
    cap = ctx.cap
    scale = (w * h) // (640 * 480)
    frames = np.stack([synth.frame(w, h, 100 + i, "rects", n_rects=800 * scale, min_size=6, max_size=32) for i in range(B)])
    d_in = torch.from_numpy(frames.reshape(-1)).cuda()
    rec = torch.zeros(B * cap * 52, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    n_pairs = (B - 1) if mm["stride"] == 1 else B // 2
    idx = torch.zeros(n_pairs * cap, dtype=torch.int32, device="cuda")
    dist = torch.zeros(n_pairs * cap, dtype=torch.int32, device="cuda")
    first = None
    for it in range(iters):
        rec.zero_()
        ctx.extract(d_in.data_ptr(), w, w * h, B, rec.data_ptr(), cnt.data_ptr(), None, s)
        ctx.match_pairs(rec.data_ptr(), cnt.data_ptr(), B, 0, mm["stride"], mm["mode"], mm["window"], mm["md"],
                        idx.data_ptr(), dist.data_ptr(), s)
        wts = torch.arange(1, rec.numel() + 1, device="cuda", dtype=torch.int64) % 65521
        sig = (int((rec.to(torch.int64) * wts).sum()), int(cnt.sum()), int((idx.to(torch.int64) * 31 + dist).sum()))
        if first is None:
            first = sig
        elif sig != first:
            bad += 1
            print("MISMATCH", name, it, sig, first)
            break
    print("%s: %d iterations, keypoints %d, matches %d, signature %s" % (name, iters, first[1], int((idx >= 0).sum()), "repeats" if bad == 0 else "CHANGED"))
sys.exit(1 if bad else 0)
