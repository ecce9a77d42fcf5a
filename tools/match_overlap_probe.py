#!/usr/bin/env python3
"""Does the matcher of step i (matrix cores) overlap the extraction of step i + 1 (vector ALUs) when the two
are issued on two streams?  The matcher touches only its own scratch in the context, so this is legal with
double-buffered records.  Prints ms per 256-frame step for 1 stream and for 2 (time.perf_counter around
200 steps, everything drained)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jetracer-orbslam2_amd"))
import orbfe  # noqa: E402
from orbfe import synth  # noqa: E402

w, h, B = 640, 480, 256
cfg = dict(levels=8, cell=8, min_arc=9, max_features=2000)
base = synth.frames(w, h, 64, first_index=1000, kind="rects", **synth.DENSE)
frames = torch.from_numpy(base).cuda()[torch.arange(B, device="cuda") % 64].contiguous()
ctx = orbfe.Context(w, h, max_batch=B, **cfg)
cap = ctx.cap
recs = [torch.zeros(B * cap * 52, dtype=torch.uint8, device="cuda") for _ in range(2)]
cnts = [torch.zeros(B, dtype=torch.int32, device="cuda") for _ in range(2)]
idx = [torch.zeros((B - 1) * cap, dtype=torch.int32, device="cuda") for _ in range(2)]
dst = [torch.zeros((B - 1) * cap, dtype=torch.int32, device="cuda") for _ in range(2)]


def run(two, steps=200):
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    if not two:
        sb = sa
    ext_done = [torch.cuda.Event() for _ in range(2)]
    mat_done = [torch.cuda.Event() for _ in range(2)]
    used = [False, False]

    def step(i):
        b = i & 1
        if two and used[b]:
            sa.wait_event(mat_done[b])  # the matcher that last read this buffer pair
        ctx.extract(frames.data_ptr(), w, w * h, B, recs[b].data_ptr(), cnts[b].data_ptr(), None, sa.cuda_stream)
        if two:
            ext_done[b].record(sa)
            sb.wait_event(ext_done[b])
        ctx.match_batch(recs[b].data_ptr(), cnts[b].data_ptr(), B, 1, -1, 256, idx[b].data_ptr(), dst[b].data_ptr(),
                        sb.cuda_stream)
        if two:
            mat_done[b].record(sb)
            used[b] = True

    for i in range(100):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return ms, int(cnts[0].sum().item()), int(idx[0].to(torch.int64).sum().item()), int(idx[1].to(torch.int64).sum().item())


for two in (False, True, False, True):
    print("%d stream(s): %.4f ms per step, checks %s" % ((2 if two else 1,) + (lambda r: (r[0], r[1:]))(run(two))))
