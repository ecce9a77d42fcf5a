#!/usr/bin/env python3
"""Time orbfe_match_batch (256-bit, all candidates) alone on random records.
usage: match_rate.py [frames] [cell] [max_features] [reps]    -> pairs/s for that shape
(640x480; cell 8 and max_features 0 gives 4800 keypoints per frame)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jetracer-orbslam2_amd"))
import orbfe  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cell = int(sys.argv[2]) if len(sys.argv) > 2 else 8
maxf = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
ctx = orbfe.Context(640, 480, max_batch=frames, cell=cell, max_features=maxf)
cap = ctx.cap
rng = np.random.default_rng(1)
rec = np.zeros((frames, cap), orbfe.KEYPOINT_DTYPE)
rec["desc"] = rng.integers(0, 256, (frames, cap, 32), dtype=np.uint8)
d_rec = torch.from_numpy(rec.view(np.uint8).reshape(-1)).cuda()
d_cnt = torch.full((frames,), cap, dtype=torch.int32, device="cuda")
d_idx = torch.zeros((frames - 1) * cap, dtype=torch.int32, device="cuda")
d_dist = torch.zeros((frames - 1) * cap, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    ctx.match_batch(d_rec.data_ptr(), d_cnt.data_ptr(), frames, 1, -1, 256, d_idx.data_ptr(), d_dist.data_ptr(), s)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ctx.match_batch(d_rec.data_ptr(), d_cnt.data_ptr(), frames, 1, -1, 256, d_idx.data_ptr(), d_dist.data_ptr(), s)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
pairs = (frames - 1) * cap * cap
print("frames %d  keypoints/frame %d  %.3f ms per call  %.2f Tpairs/s" % (frames, cap, ms, pairs / ms / 1e9))
