"""Deterministic synthetic inputs for tests and bench.

There is no camera, dataset or network here, so frames are generated: a counter-based
splitmix64 stream seeded with 0x0B5EED00 + frame_index paints filled rectangles on mid-gray
and adds +-3 uniform noise.  Pure numpy; no dependency on the oracle or on the HIP library.
"""
import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
BASE_SEED = 0x0B5EED00


def splitmix64(seed, n, start=0):
    """Outputs start .. start+n-1 of the splitmix64 stream whose state starts at `seed`."""
    with np.errstate(over="ignore"):
        i = np.arange(start + 1, start + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


# corner-rich scene: 2400-2570 FAST-9 keypoints per 640x480 frame with 8-px cells over 4 levels
# (about 8 % of the level-0 pixels are FAST-9 corners), i.e. just above a 2000-feature budget
DENSE = dict(n_rects=800, min_size=6, max_size=32)


def frame(width, height, index=0, kind="rects", n_rects=96, min_size=6, max_size=None):
    """One u8 grayscale frame [height, width].

    kind: 'rects' (n_rects filled rectangles + noise; sides uniform in min_size..max_size,
          max_size None = a fifth of the frame as in SURVEY.md 8d), 'uniform' (i.i.d. bytes),
          'const' (128 everywhere), 'checker' (period-16 checkerboard 64/192).
    `**DENSE` gives a corner-rich scene for the 2000-features/frame bench configuration.
    """
    seed = BASE_SEED + index
    if kind == "const":
        return np.full((height, width), 128, dtype=np.uint8)
    if kind == "checker":
        yy, xx = np.mgrid[0:height, 0:width]
        return np.where(((xx // 8) + (yy // 8)) % 2 == 0, 64, 192).astype(np.uint8)
    if kind == "uniform":
        return (splitmix64(seed, width * height) & np.uint64(255)).astype(np.uint8).reshape(
            height, width)
    if kind != "rects":
        raise ValueError(kind)
    img = np.full((height, width), 128, dtype=np.int16)
    r = splitmix64(seed, 5 * n_rects)
    max_w = max(width // 5, min_size) if max_size is None else max_size
    max_h = max(height // 5, min_size) if max_size is None else max_size
    for k in range(n_rects):
        x0 = int(r[5 * k] % np.uint64(width))
        y0 = int(r[5 * k + 1] % np.uint64(height))
        rw = min_size + int(r[5 * k + 2] % np.uint64(max_w - min_size + 1))
        rh = min_size + int(r[5 * k + 3] % np.uint64(max_h - min_size + 1))
        g = int(r[5 * k + 4] % np.uint64(256))
        img[y0:min(y0 + rh, height), x0:min(x0 + rw, width)] = g
    noise = (splitmix64(seed, width * height, start=5 * n_rects) % np.uint64(7)).astype(
        np.int16) - 3
    img += noise.reshape(height, width)
    return np.clip(img, 0, 255).astype(np.uint8)


def frames(width, height, count, first_index=0, kind="rects", **kw):
    return np.stack([frame(width, height, first_index + i, kind, **kw) for i in range(count)])


def shifted_pair(width, height, index=0, dx=3, dy=0, **kw):
    """A frame and the same scene shifted by (dx, dy) with fresh +-2 noise (config C3)."""
    a = frame(width, height, index, "rects", **kw)
    b = np.roll(a, (dy, dx), axis=(0, 1)).astype(np.int16)
    noise = (splitmix64(BASE_SEED + 0x1000 + index, width * height) % np.uint64(5)).astype(
        np.int16) - 2
    b = np.clip(b + noise.reshape(height, width), 0, 255).astype(np.uint8)
    return a, b


def descriptors(n, seed=1):
    """n random 256-bit descriptors as u8 [n, 32]."""
    return splitmix64(BASE_SEED ^ (seed * 0x10001), n * 4).view(np.uint8).reshape(n, 32).copy()
