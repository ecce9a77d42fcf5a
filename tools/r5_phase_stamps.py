#!/usr/bin/env python3
"""round 5 diagnostic: where a detect / describe workgroup's lifetime goes, phase by phase (s_memtime ticks of wave 0 between the
kernel's barriers, averaged over every workgroup of a few default bench steps).  Needs ORBFE_LIB=.../.variants/phstamps/liborbfe.so
(tools/r5_phase_stamps_build.sh)."""
import contextlib
import ctypes
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "jetracer-orbslam2_amd"))
import orbfe  # noqa: E402

sys.argv = ["bench.py", "--no-cpu-baseline", "--no-extras", "--steps", "5", "--warmup", "2"] + sys.argv[1:]
import bench  # noqa: E402

buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
out = (ctypes.c_ulonglong * 32)()
lib = orbfe.lib()
lib.orbfe_debug_phase.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
assert lib.orbfe_debug_phase(out, 1) == 0
v = [int(x) for x in out]
for name, o, phases, unit in (("detect_tile_kernel", 0, ("A tile load", "B compass + compaction", "C ring test", "D 3x3 maximum + cell keys"), "ring-test candidates"),
                              ("describe_tile_kernel", 8, ("stage: tile DMA + keypoint list", "A moments", "B angles", "C descriptors"), "keypoints")):
    n = v[o + 4]
    if not n:
        continue
    life = v[o + 7] / n
    print("%s: %d workgroups, lifetime %.0f ticks = %.2f us (core clock %.0f MHz), %.1f %s per workgroup" %
          (name, n, life, v[o + 5] / n / 100.0, v[o + 7] / max(v[o + 5], 1) * 100.0, v[o + 6] / n, unit))
    for i, ph in enumerate(phases):
        print("    %-34s %8.0f ticks  %5.1f %%" % (ph, v[o + i] / n, 100.0 * v[o + i] / max(v[o + 7], 1)))
