#!/usr/bin/env python3
"""round 5: orbfe_match_batch (256-bit, all candidates) in both matrix-core forms over call shapes -- where the size rule
(match_mfma_uses_tile) should switch.  Prints ms per call for ORBFE_MATCH=stream and =tile and the (pair, 512-query tile) item count."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jetracer-orbslam2_amd"))
import orbfe  # noqa: E402

rng = np.random.default_rng(1)
s = torch.cuda.current_stream().cuda_stream
print("pairs keypoints items | stream ms  tile ms | tile / stream")
for n in (405, 1000, 2000, 4800, 8192):
    for pairs in (1, 2, 4, 8, 15, 32, 64, 128, 255):
        if pairs * n * n > 255 * 2000 * 2000 * 2: continue
        frames = pairs + 1
        rec = np.zeros((frames, n), orbfe.KEYPOINT_DTYPE)
        rec["desc"] = rng.integers(0, 256, (frames, n, 32), dtype=np.uint8)
        d_rec = torch.from_numpy(rec.view(np.uint8).reshape(-1)).cuda()
        d_cnt = torch.full((frames,), n, dtype=torch.int32, device="cuda")
        d_idx = torch.zeros(pairs * n, dtype=torch.int32, device="cuda")
        ms = {}
        for form in ("stream", "tile"):
            os.environ["ORBFE_MATCH"] = form
            ctx = orbfe.Context(1280, 720, levels=1, cell=8, min_arc=9, max_features=n, max_batch=frames)
            assert ctx.cap == n
            for _ in range(3): ctx.match_batch(d_rec.data_ptr(), d_cnt.data_ptr(), frames, 1, -1, 256, d_idx.data_ptr(), None, s)
            reps = 50 if pairs * n * n < 4e8 else 10
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): ctx.match_batch(d_rec.data_ptr(), d_cnt.data_ptr(), frames, 1, -1, 256, d_idx.data_ptr(), None, s)
            e1.record()
            torch.cuda.synchronize()
            ms[form] = e0.elapsed_time(e1) / reps
            del ctx
        items = pairs * ((n + 511) // 512)
        print("%5d %9d %5d | %8.4f %8.4f | %.2f" % (pairs, n, items, ms["stream"], ms["tile"], ms["tile"] / ms["stream"]), flush=True)
