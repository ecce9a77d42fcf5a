"""The invariant behind detect_tile_kernel's unconditional tile loads, checked on the host for many geometries.

Round 3 removed the range tests from the detection tile loads and gave the context's pyramid guard bands instead.  A
work-in-progress build of that change faulted on the GPU ("Memory access fault by GPU node-2", gpurun_out/r3i: the
bands were not allocated yet).  orbfe_layout_bounds (host code, no device) reports, for a configuration, the lowest and
highest byte any tile load can touch; orbfe_create refuses a geometry that violates the bound, and this test walks
random geometries so that a change of the tile size, the level packing or the pitch alignment cannot bring the fault
back silently (there is no GPU sanitizer on this pool).  No GPU."""
import ctypes as C

import numpy as np
import pytest


def bounds(orbfe, w, h, levels, cell, max_batch):
    cfg = orbfe.Config(w, h, levels, cell, 13, 9, 0, 0, max_batch, 0, 0)
    lo, hi, pyr, guard, nt = C.c_longlong(), C.c_longlong(), C.c_ulonglong(), C.c_ulonglong(), C.c_int()
    rc = orbfe.lib().orbfe_layout_bounds(C.byref(cfg), C.byref(lo), C.byref(hi), C.byref(pyr), C.byref(guard), C.byref(nt))
    return rc, lo.value, hi.value, pyr.value, guard.value, nt.value


def restated_extent(w, h, levels, cell, max_batch):
    """The same quantity from the layout rules of DESIGN.md section 3, written independently: level pitch = width
    rounded up to 64, level offsets 256-aligned, 64 x 64 tiles on levels with cell >> l >= 1 and at least 7 x 7 pixels;
    a tile reads rows y0 - 4 .. y0 + 67 and bytes x0 - 4 .. x0 + 67."""
    off, lv = 0, []
    for l in range(levels):
        wl, hl = w >> l, h >> l
        p = -(-max(wl, 1) // 64) * 64
        lv.append((wl, hl, p, off))
        off += -(-(p * max(hl, 1)) // 256) * 256
    stride = off
    lo, hi, n = None, None, 0
    for l, (wl, hl, p, o) in enumerate(lv):
        if (cell >> l) == 0 or wl == 0 or hl == 0:
            break
        if wl < 7 or hl < 7:
            continue
        for ty in range(-(-hl // 64)):
            for tx in range(-(-wl // 64)):
                first = o + (64 * ty - 4) * p + 64 * tx - 4
                last = o + (64 * ty + 67) * p + 64 * tx + 67 + (max_batch - 1) * stride
                lo = first if lo is None else min(lo, first)
                hi = last if hi is None else max(hi, last)
                n += 1
    return lo, hi, stride * max_batch, n


def test_every_tile_load_lies_inside_the_allocation():
    import orbfe
    rng = np.random.default_rng(20261005)
    cases = [(640, 480, 8, 8, 256), (848, 480, 8, 8, 128), (1280, 720, 8, 8, 64), (3840, 2160, 12, 32, 1),
             (16384, 16384, 16, 64, 1), (8, 8, 1, 8, 1), (8, 8, 16, 64, 4096), (63, 65, 7, 16, 3), (64, 64, 2, 32, 1),
             (65, 8, 3, 8, 2), (8, 16384, 5, 8, 1), (16384, 8, 4, 64, 7)]
    for _ in range(400):
        cases.append((int(rng.integers(8, 16385)) if rng.random() < 0.2 else int(rng.integers(8, 700)),
                      int(rng.integers(8, 16385)) if rng.random() < 0.2 else int(rng.integers(8, 700)),
                      int(rng.integers(1, 17)), int(rng.choice([8, 16, 32, 64])),
                      int(rng.choice([1, 2, 7, 8, 9, 256, 4096]))))
    seen_tiles = 0
    for w, h, levels, cell, mb in cases:
        rc, lo, hi, pyr, guard, nt = bounds(orbfe, w, h, levels, cell, mb)
        assert rc == 0, (w, h, levels, cell, mb)
        rlo, rhi, rpyr, rn = restated_extent(w, h, levels, cell, mb)
        assert pyr == rpyr and nt == rn, (w, h, levels, cell, mb)
        if nt == 0:
            continue
        assert (lo, hi) == (rlo, rhi), (w, h, levels, cell, mb)
        assert -guard <= lo, "a tile load reaches %d bytes before the allocation: %r" % (-guard - lo, (w, h, levels, cell, mb))
        assert hi < pyr + guard, "a tile load reaches %d bytes past the allocation: %r" % (hi - pyr - guard + 1, (w, h, levels, cell, mb))
        assert lo < 0, "the first tile's halo always starts above the image: the front band is needed"
        seen_tiles += nt
    assert seen_tiles > 100000


def test_the_guard_bands_are_not_oversized_by_luck():
    """The bound is tight to within the rounding: for a frame whose last detection level's last tile hangs over the
    end, what is needed after the pyramid stays below the band but above a fifth of it (a change that halves the
    band would fail the test above on such a geometry)."""
    import orbfe
    worst = 0.0
    for w, h, levels, cell in [(640, 480, 1, 32), (65, 65, 1, 8), (129, 65, 1, 8), (16384, 65, 1, 64), (4097, 129, 1, 32)]:
        rc, lo, hi, pyr, guard, nt = bounds(orbfe, w, h, levels, cell, 1)
        assert rc == 0 and nt > 0
        worst = max(worst, (hi - pyr + 1) / guard)
        assert -lo <= guard
    assert 0.2 < worst <= 1.0


def test_invalid_configurations_are_refused():
    import orbfe
    assert bounds(orbfe, 4, 480, 1, 32, 1)[0] == orbfe.ERR_INVALID_ARG
    assert bounds(orbfe, 640, 480, 17, 32, 1)[0] == orbfe.ERR_INVALID_ARG
    assert bounds(orbfe, 640, 480, 1, 12, 1)[0] == orbfe.ERR_INVALID_ARG
    assert bounds(orbfe, 640, 480, 1, 32, 0)[0] == orbfe.ERR_INVALID_ARG
