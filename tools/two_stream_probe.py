#!/usr/bin/env python3
"""Does issuing the two halves of a 256-frame batch on two streams (so that one half's VALU-bound
detection overlaps the other half's latency-bound description) beat one stream?  Same work, same
results: extraction per half, then one match over the whole batch.  Prints ms per 256-frame step."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jetracer-orbslam2_amd"))
import orbfe  # noqa: E402
from orbfe import synth  # noqa: E402

w, h, B = 640, 480, 256
cfg = dict(levels=8, cell=8, min_arc=9, max_features=2000)
base = synth.frames(w, h, 16, first_index=1000, kind="rects", **synth.DENSE)
frames = torch.from_numpy(base).cuda()[torch.arange(B, device="cuda") % 16].contiguous()


def run(n_streams, reps=20):
    per = B // n_streams
    ctxs = [orbfe.Context(w, h, max_batch=B if i == 0 else per, **cfg) for i in range(n_streams)]
    cap = ctxs[0].cap
    rec = torch.zeros(B * cap * 52, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    idx = torch.zeros((B - 1) * cap, dtype=torch.int32, device="cuda")
    dst = torch.zeros((B - 1) * cap, dtype=torch.int32, device="cuda")
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    done = [torch.cuda.Event() for _ in range(n_streams)]

    def step():
        for i, (c, st) in enumerate(zip(ctxs, streams)):
            c.extract(frames.data_ptr() + i * per * w * h, w, w * h, per, rec.data_ptr() + i * per * cap * 52,
                      cnt.data_ptr() + i * per * 4, None, st.cuda_stream)
            done[i].record(st)
        for i in range(1, n_streams):
            streams[0].wait_event(done[i])
        ctxs[0].match_batch(rec.data_ptr(), cnt.data_ptr(), B, 1, -1, 256, idx.data_ptr(), dst.data_ptr(),
                            streams[0].cuda_stream)
        for i in range(1, n_streams):  # the next step's extraction must not overwrite records being matched
            e = torch.cuda.Event()
            e.record(streams[0])
            streams[i].wait_event(e)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(streams[0])
    for _ in range(reps):
        step()
    e1.record(streams[0])
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    chk = int(cnt.sum().item()), int(idx.to(torch.int64).sum().item())
    for c in ctxs:
        c.close()
    return ms, chk


for n in (1, 2, 4, 1, 2):
    ms, chk = run(n)
    print("%d stream(s): %.3f ms per 256-frame step   checksum %s" % (n, ms, chk))
