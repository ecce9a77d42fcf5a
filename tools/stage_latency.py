#!/usr/bin/env python3
"""Latency of the stage API for one frame (the call sequence of buildStream.cpp:424-460) on cuda:0.
usage: stage_latency.py [width height levels]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jetracer-orbslam2_amd"))
import torch
import orbfe
from orbfe import synth

w, h, levels = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (640, 480, 1)
L = orbfe.lib()
s = torch.cuda.current_stream().cuda_stream
k = ((w + 31) // 32) * ((h + 31) // 32)
gray = torch.from_numpy(synth.frame(w, h, 1, "rects", **synth.DENSE)).cuda()
imgs = [torch.zeros((h >> l, w >> l), dtype=torch.uint8, device="cuda") for l in range(levels)]
resp = [torch.zeros((h >> l, w >> l), dtype=torch.float32, device="cuda") for l in range(levels)]
lv = orbfe.make_levels([(imgs[l].data_ptr(), w >> l, h >> l, w >> l) for l in range(levels)],
                       [(resp[l].data_ptr(), w >> l, h >> l, (w >> l) * 4) for l in range(levels)])
lut = torch.zeros(65536, dtype=torch.uint8, device="cuda")
grid = torch.zeros(4 * k, dtype=torch.float32, device="cuda")
angle = torch.zeros(k, dtype=torch.float32, device="cuda")
desc = torch.zeros(k * 32, dtype=torch.uint8, device="cuda")
d32 = torch.zeros(k, dtype=torch.int32, device="cuda")
L.orbfe_fast_calculate_lut(lut.data_ptr(), 12, s)
b = grid.data_ptr()


def frame():
    L.orbfe_gaussian_blur_3x3(imgs[0].data_ptr(), w, gray.data_ptr(), w, w, h, s)
    L.orbfe_pyramid_create_levels(lv, levels, s)
    L.orbfe_detect(lv, levels, lut.data_ptr(), 13.0, b, b + 8 * k, b + 12 * k, s)
    L.orbfe_compute_fast_angle(angle.data_ptr(), b, imgs[0].data_ptr(), w, w, h, k, s)
    L.orbfe_calc_orb(angle.data_ptr(), b, desc.data_ptr(), d32.data_ptr(), imgs[0].data_ptr(), w, w, h, k, s)


for _ in range(20):
    frame()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 200
e0.record()
for _ in range(n):
    frame()
e1.record()
torch.cuda.synchronize()
print("stage API, %dx%d, %d level(s), K = %d: %.1f us per frame (%d keypoints)"
      % (w, h, levels, k, e0.elapsed_time(e1) / n * 1e3, int((grid[2 * k:3 * k] > 0).sum())))

# the same frame through the batch API with a batch of one (orbfe_extract = 4 launches), and the
# 8-level / 2000-feature regime of the bench
for tag, cfg in (("reference regime", dict(levels=levels, cell=32, min_arc=12)),
                 ("bench regime", dict(levels=8, cell=8, min_arc=9, max_features=2000))):
    ctx = orbfe.Context(w, h, max_batch=1, **cfg)
    rec = torch.zeros(ctx.cap * 52, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    for _ in range(20):
        ctx.extract(gray.data_ptr(), w, w * h, 1, rec.data_ptr(), cnt.data_ptr(), None, s)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        ctx.extract(gray.data_ptr(), w, w * h, 1, rec.data_ptr(), cnt.data_ptr(), None, s)
    e1.record()
    torch.cuda.synchronize()
    print("batch API, batch of 1, %s: %.1f us per frame (%d keypoints)"
          % (tag, e0.elapsed_time(e1) / n * 1e3, int(cnt.item())))
