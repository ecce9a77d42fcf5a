#!/usr/bin/env python3
"""Are the stages' times in the bench loop independent of their neighbours?  4096-frame C2 steps, HIP events around every
stage, three loops on one stream: the whole step; the step without the matcher; detect alone (pyramid once).  If a stage
runs faster when a (power-hungry) neighbour is absent, stage times are coupled through the chip's power / clock management
and A/B decisions must be taken on the whole step."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jetracer-orbslam2_amd"))
import orbfe  # noqa: E402
from orbfe import synth  # noqa: E402

w, h, B = 640, 480, 4096
cfg = dict(levels=8, cell=8, min_arc=9, max_features=2000)
base = synth.frames(w, h, 64, first_index=1000, kind="rects", **synth.DENSE)
frames = torch.from_numpy(base).cuda()[torch.arange(B, device="cuda") % 64].contiguous()
ctx = orbfe.Context(w, h, max_batch=B, **cfg)
cap = ctx.cap
rec = torch.zeros(B * cap * 52, dtype=torch.uint8, device="cuda")
cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
idx = torch.zeros((B - 1) * cap, dtype=torch.int32, device="cuda")
dst = torch.zeros((B - 1) * cap, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
stages = {"pyramid": lambda: ctx.build_pyramid(frames.data_ptr(), w, w * h, B, s),
          "detect": lambda: ctx.detect_batch(B, s),
          "describe": lambda: ctx.describe_batch(B, rec.data_ptr(), cnt.data_ptr(), None, s),
          "match": lambda: ctx.match_batch(rec.data_ptr(), cnt.data_ptr(), B, 1, -1, 256, idx.data_ptr(), dst.data_ptr(), s)}


def loop(names, steps=60, warm=15):
    for _ in range(warm):
        for n in names:
            stages[n]()
    torch.cuda.synchronize()
    acc = {n: 0.0 for n in names}
    evs = []
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(steps):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        e[0].record()
        for i, n in enumerate(names):
            stages[n]()
            e[i + 1].record()
        evs.append(e)
    t1.record()
    torch.cuda.synchronize()
    for e in evs:
        for i, n in enumerate(names):
            acc[n] += e[i].elapsed_time(e[i + 1])
    return t0.elapsed_time(t1) / steps, {n: round(v / steps, 4) for n, v in acc.items()}


for names in (["pyramid", "detect", "describe", "match"], ["pyramid", "detect", "describe"], ["detect"], ["match"],
              ["pyramid", "detect", "describe", "match"], ["detect"], ["detect", "match"]):
    ms, per = loop(names)
    print("%-40s %.4f ms per step  %s" % (" + ".join(names), ms, per))
