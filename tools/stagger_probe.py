#!/usr/bin/env python3
"""Does the HBM-bound pyramid of one half batch hide under the VALU-bound detection of the other when the two halves
run on two streams STAGGERED by one stage (option 1)?  Two contexts, two streams:
    stream 1:  pyramid(A)  detect(A)  describe(A)                       match(all)
    stream 2:      wait -> pyramid(B)  detect(B)  describe(B)  -> join
against everything on one stream.  A detect workgroup needs 26 KB of LDS and 4 wave slots, six fit a CU, which leaves 8
wave slots and 4 KB for two pyramid workgroups.  Same results either way (checksums printed).
usage: stagger_probe.py [frames per step, default 2048]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jetracer-orbslam2_amd"))
import orbfe  # noqa: E402
from orbfe import synth  # noqa: E402

w, h = 640, 480
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
cfg = dict(levels=8, cell=8, min_arc=9, max_features=2000)
base = synth.frames(w, h, 32, first_index=1000, kind="rects", **synth.DENSE)
frames = torch.from_numpy(base).cuda()[torch.arange(B, device="cuda") % 32].contiguous()


def run(mode, reps=20):
    half = B // 2
    ctxs = [orbfe.Context(w, h, max_batch=B if mode == "one" else half, **cfg) for _ in range(1 if mode == "one" else 2)]
    cap = ctxs[0].cap
    rec = torch.zeros(B * cap * 52, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(B, dtype=torch.int32, device="cuda")
    idx = torch.zeros((B - 1) * cap, dtype=torch.int32, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    matcher = ctxs[0] if mode == "one" else orbfe.Context(w, h, max_batch=B, **cfg)

    def part(c, st, first, n, ev_after_pyramid=None):
        c.build_pyramid(frames.data_ptr() + first * w * h, w, w * h, n, st.cuda_stream)
        if ev_after_pyramid is not None:
            ev_after_pyramid.record(st)
        c.detect_batch(n, st.cuda_stream)
        c.describe_batch(n, rec.data_ptr() + first * cap * 52, cnt.data_ptr() + first * 4, None, st.cuda_stream)

    def step():
        if mode == "one":
            part(ctxs[0], s1, 0, B)
        else:
            e = torch.cuda.Event()
            part(ctxs[0], s1, 0, half, e)
            if mode == "stagger":
                s2.wait_event(e)  # pyramid(B) starts when pyramid(A) is done: it runs under detect(A)
            part(ctxs[1], s2, half, half)
            done = torch.cuda.Event()
            done.record(s2)
            s1.wait_event(done)
        matcher.match_batch(rec.data_ptr(), cnt.data_ptr(), B, 1, -1, 256, idx.data_ptr(), None, s1.cuda_stream)
        if mode != "one":  # the next step's half B must not overwrite records being matched
            m = torch.cuda.Event()
            m.record(s1)
            s2.wait_event(m)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s1)
    for _ in range(reps):
        step()
    e1.record(s1)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    chk = (int(cnt.sum().item()), int(idx.to(torch.int64).sum().item()), int(rec.to(torch.int64).sum().item()))
    for c in ctxs:
        c.close()
    if mode != "one":
        matcher.close()
    return ms, chk


for mode in ("one", "two", "stagger", "one", "stagger"):
    ms, chk = run(mode)
    print("%-8s %.4f ms per %d-frame step (%.4f per 256)   checks %s" % (mode, ms, B, ms * 256 / B, chk))
