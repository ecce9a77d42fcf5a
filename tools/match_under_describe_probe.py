#!/usr/bin/env python3
"""Where in the next step should the matcher of step i run?  Two streams, one context (the matcher touches only its own
scratch): stream A issues pyramid, detect, describe of step i + 1; stream B issues match(i) after an event on stream A:
  'extract'  -- right after extract(i) (tools/match_overlap_probe.py: under the pyramid and the head of detect)
  'detect'   -- after detect(i + 1): under the latency-bound describe kernel
  'pyramid'  -- after the pyramid of step i + 1: under detect
Prints ms per 256-frame step (and for 1024 frames) for one stream and each placement."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jetracer-orbslam2_amd"))
import orbfe  # noqa: E402
from orbfe import synth  # noqa: E402

w, h = 640, 480
cfg = dict(levels=8, cell=8, min_arc=9, max_features=2000)
base = synth.frames(w, h, 64, first_index=1000, kind="rects", **synth.DENSE)


def bench(B, steps):
    frames = torch.from_numpy(base).cuda()[torch.arange(B, device="cuda") % 64].contiguous()
    ctx = orbfe.Context(w, h, max_batch=B, **cfg)
    cap = ctx.cap
    recs = [torch.zeros(B * cap * 52, dtype=torch.uint8, device="cuda") for _ in range(2)]
    cnts = [torch.zeros(B, dtype=torch.int32, device="cuda") for _ in range(2)]
    idx = [torch.zeros((B - 1) * cap, dtype=torch.int32, device="cuda") for _ in range(2)]
    dst = [torch.zeros((B - 1) * cap, dtype=torch.int32, device="cuda") for _ in range(2)]

    def run(place):
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        if place is None:
            sb = sa
        mat_done = [torch.cuda.Event() for _ in range(2)]
        used = [False, False]
        pending = [None]  # (buffer index) whose match has not been issued yet

        def issue_match(b):
            ctx.match_batch(recs[b].data_ptr(), cnts[b].data_ptr(), B, 1, -1, 256, idx[b].data_ptr(), dst[b].data_ptr(), sb.cuda_stream)
            if place is not None:
                mat_done[b].record(sb)
                used[b] = True

        def hook(where):
            if place == where and pending[0] is not None:
                ev = torch.cuda.Event()
                ev.record(sa)
                sb.wait_event(ev)
                issue_match(pending[0])
                pending[0] = None

        def step(i):
            b = i & 1
            if place is not None and used[b]:
                sa.wait_event(mat_done[b])
            ctx.build_pyramid(frames.data_ptr(), w, w * h, B, sa.cuda_stream)
            hook("pyramid")
            ctx.detect_batch(B, sa.cuda_stream)
            hook("detect")
            ctx.describe_batch(B, recs[b].data_ptr(), cnts[b].data_ptr(), None, sa.cuda_stream)
            if place is None:
                issue_match(b)
            elif place == "extract":
                ev = torch.cuda.Event()
                ev.record(sa)
                sb.wait_event(ev)
                issue_match(b)
            else:
                # the previous step's match must have been issued by a hook; this step's waits for the next step's hook
                assert pending[0] is None
                pending[0] = b

        for i in range(30):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        if pending[0] is not None:
            ev = torch.cuda.Event()
            ev.record(sa)
            sb.wait_event(ev)
            issue_match(pending[0])
            pending[0] = None
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, int(idx[0].to(torch.int64).sum().item()), int(idx[1].to(torch.int64).sum().item())

    for place in (None, "extract", "pyramid", "detect", None, "detect", "pyramid"):
        ms, c0, c1 = run(place)
        print("B = %4d  matcher %-28s %.4f ms per step  (checks %d %d)" % (B, "on the same stream" if place is None else "after %s, 2nd stream" % place, ms, c0, c1))
    ctx.close()


bench(256, 200)
bench(1024, 60)
bench(4096, 20)
