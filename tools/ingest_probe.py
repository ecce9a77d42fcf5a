"""Where does a small slot's time go?  Runs the staging ring (include/orbfe_ingest.h) at several slot sizes / ring depths
and prints, per configuration: pipelined ms per slot, the ring's own event durations, and the host time spent inside
submit() and wait().  python tools/ingest_probe.py [frames_per_slot ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jetracer-orbslam2_amd"))
import numpy as np
import torch
import orbfe
from orbfe import synth

EXT = dict(levels=8, cell=8, min_arc=9, max_features=2000)


def run(F, slots, passes, match=True):
    w, h = 640, 480
    ctx = orbfe.Context(w, h, max_batch=F, **EXT)
    ing = orbfe.Ingest(ctx, F, slots=slots, match_mode=1 if match else -1, download_matches=2 if match else 0)
    base = synth.frames(w, h, min(F, 16), first_index=7, kind="rects", **synth.DENSE)
    for s in range(slots):
        ing.host_frames(s)[:] = base[np.arange(F) % len(base)]
    for s in range(slots):
        ing.submit(s, F)
    for s in range(slots):
        ing.wait(s)
    t_sub = t_wait = 0.0
    t0 = time.perf_counter()
    for i in range(passes):
        s = i % slots
        if i >= slots:
            a = time.perf_counter()
            ing.wait(s)
            t_wait += time.perf_counter() - a
        a = time.perf_counter()
        ing.submit(s, F)
        t_sub += time.perf_counter() - a
    for i in range(passes, passes + slots):
        a = time.perf_counter()
        ing.wait(i % slots)
        t_wait += time.perf_counter() - a
    total = time.perf_counter() - t0
    tm = [ing.timing(s) for s in range(slots)]
    up = sum(t["upload_ms"] for t in tm) / slots
    cm = sum(t["compute_ms"] for t in tm) / slots
    dn = sum(t["download_ms"] for t in tm) / slots
    print("F=%5d slots=%d passes=%3d match=%d: %.3f ms/slot (%.0f frames/s) | events: up %.3f cmp %.3f down %.3f | host: submit %.3f wait %.3f ms/slot"
          % (F, slots, passes, match, total / passes * 1e3, F * passes / total, up, cm, dn, t_sub / passes * 1e3, t_wait / passes * 1e3), flush=True)
    ing.close()
    ctx.close()


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [64, 256, 1024]
    for F in sizes:
        for slots in (2, 3, 4, 6):
            run(F, slots, max(24, 8 * slots))
        run(F, 3, 48, match=False)
