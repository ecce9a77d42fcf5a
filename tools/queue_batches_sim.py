#!/usr/bin/env python3
"""How many 64-candidate ring-test batches does detect_tile_kernel run per wave?  CPU-side count on the oracle's
pyramids of bench scenes (compass pre-test of arcs 9..11 in numpy): per-wave queues (rounds 1-2: every wave rounds
its own count up) against one queue per workgroup (round 3) against the work itself.  DESIGN.md 4.2 quotes it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "jetracer-orbslam2_amd"))
import oracle  # noqa: E402
from orbfe import synth  # noqa: E402

w, h, t = 640, 480, 13
frames = synth.frames(w, h, 4, first_index=1000, kind="rects", **synth.DENSE)
cfg = oracle.make_config(w, h, levels=8, cell=8, min_arc=9, max_features=2000)
per_wave = ideal = shared = 0.0
n_waves = 0
counts = []
for f in range(len(frames)):
    ref = oracle.extract_frame(frames[f], cfg, want_pyramid=True)
    for lvl in range(4):
        img = ref["pyramid"][lvl].astype(np.int32)
        H, W = img.shape
        c = img[3:H - 3, 3:W - 3]
        N, S, E, Wp = img[0:H - 6, 3:W - 3], img[6:H, 3:W - 3], img[3:H - 3, 6:W], img[3:H - 3, 0:W - 6]
        br = ((N > c + t) | (S > c + t)) & ((E > c + t) | (Wp > c + t))
        dk = ((N < c - t) | (S < c - t)) & ((E < c - t) | (Wp < c - t))
        cand = np.zeros((H, W), bool)
        cand[3:H - 3, 3:W - 3] = br | dk
        for ty in range((H + 63) // 64):
            for tx in range((W + 63) // 64):
                y0, x0 = ty * 64, tx * 64
                waves = [0, 0, 0, 0]
                for r in range(66):  # score-tile rows / columns 0..65 <-> image y0 - 1 + r, x0 - 1 + px
                    y = y0 - 1 + r
                    if y < 0 or y >= H:
                        continue
                    for px in range(66):
                        x = x0 - 1 + px
                        if x < 0 or x >= W or not cand[y, x]:
                            continue
                        if 1 <= r <= 64 and 1 <= px <= 64:
                            wv = (r - 1) // 16           # a wave owns a band of 16 rows
                        elif (r == 0 or r == 65) and 1 <= px <= 64:
                            wv = 0                       # halo trip: lanes 0..31
                        else:
                            wv = ((32 + r) if px == 0 else (32 + 66 + r)) // 64  # halo columns: lanes 32..163
                        waves[wv] += 1
                for n in waves:
                    per_wave += -(-n // 64)
                    ideal += n / 64
                    counts.append(n)
                shared += -(-sum(waves) // 64)
                n_waves += 4
print("waves %d, candidates per wave %.1f; batches per wave: per-wave queues %.2f, one queue per workgroup %.2f, the work %.2f"
      % (n_waves, np.mean(counts), per_wave / n_waves, shared / n_waves, ideal / n_waves))
