#!/usr/bin/env python3
"""Single-frame latency of the tracking loop a camera pipeline runs (one frame in, keypoints + matches
against the previous frame out), eager launches against one hipGraph replay per frame.
usage: latency_probe.py [ref|c2] [iters]
  ref: 640x480, 6 levels, cell 32, FAST-12, <= 300 keypoints, 32-bit windowed match (the reference's regime)
  c2 : 640x480, 8 levels, cell 8, FAST-9, top-2000, 256-bit brute-force match
Prints microseconds per frame: host-synchronised after every frame (latency) and back to back."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jetracer-orbslam2_amd"))
import orbfe  # noqa: E402
from orbfe import synth  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "ref"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
w, h = 640, 480
if mode == "ref":
    ctx = orbfe.Context(w, h, max_batch=2)
    mm = dict(mode=0, window=32, max_distance=8)
else:
    ctx = orbfe.Context(w, h, max_batch=2, levels=8, cell=8, min_arc=9, max_features=2000)
    mm = dict(mode=1, window=-1, max_distance=64)
cap = ctx.cap
frames = [torch.from_numpy(synth.frame(w, h, i, "rects", n_rects=800, min_size=6, max_size=32)).cuda() for i in range(8)]
d_in = torch.zeros(h * w, dtype=torch.uint8, device="cuda")
rec = torch.zeros(2 * cap * 52, dtype=torch.uint8, device="cuda")
cnt = torch.zeros(2, dtype=torch.int32, device="cuda")
idx = torch.zeros(cap, dtype=torch.int32, device="cuda")
side = torch.cuda.Stream()


def frame_step(s):
    # slot 0 = previous frame, slot 1 = current frame
    ctx.extract(d_in.data_ptr(), w, w * h, 1, rec.data_ptr() + cap * 52, cnt.data_ptr() + 4, None, s)
    ctx.match_pairs(rec.data_ptr(), cnt.data_ptr(), 2, 0, 1, mm["mode"], mm["window"], mm["max_distance"], idx.data_ptr(), None, s)
    rec[:cap * 52].copy_(rec[cap * 52:], non_blocking=True)
    cnt[:1].copy_(cnt[1:], non_blocking=True)


def run(step, sync_each):
    with torch.cuda.stream(side):
        for i in range(20):
            d_in.copy_(frames[i % 8].reshape(-1), non_blocking=True)
            step()
        side.synchronize()
        t0 = time.perf_counter()
        for i in range(iters):
            d_in.copy_(frames[i % 8].reshape(-1), non_blocking=True)
            step()
            if sync_each:
                side.synchronize()
        side.synchronize()
        return (time.perf_counter() - t0) / iters * 1e6


with torch.cuda.stream(side):
    eager = lambda: frame_step(side.cuda_stream)
    e_lat, e_thr = run(eager, True), run(eager, False)
    g = torch.cuda.CUDAGraph()
    side.synchronize()
    with torch.cuda.graph(g, stream=side):
        frame_step(side.cuda_stream)
    g_lat, g_thr = run(g.replay, True), run(g.replay, False)
print("mode %s  keypoints %d  eager: %.1f us/frame synchronised, %.1f back to back   hipGraph: %.1f synchronised, %.1f back to back"
      % (mode, int(cnt.cpu()[0]), e_lat, e_thr, g_lat, g_thr))
