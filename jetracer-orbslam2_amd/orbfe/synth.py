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


# ---- depth frames and camera rigs for align_depth_to_other --------------------------------
def depth_frame(width, height, index=0, n_rects=40, holes=0.15, near=300, far=4000):
    """One u16 depth frame [height, width] in the units of a D4xx (1 unit = depth_scale metres, normally 1 mm):
    a slanted background wall `far` units away with +-2 units of noise, n_rects fronto-parallel boxes between
    `near` and `far`, and a fraction `holes` of invalid pixels (0), half of them as 1..12-pixel horizontal runs
    (the shadow bands a stereo depth camera leaves beside foreground edges)."""
    seed = BASE_SEED + 0x2000 + index
    yy, xx = np.mgrid[0:height, 0:width]
    img = (far - 0.6 * xx - 0.25 * yy).astype(np.int32)
    r = splitmix64(seed, 5 * n_rects)
    for k in range(n_rects):
        x0 = int(r[5 * k] % np.uint64(width))
        y0 = int(r[5 * k + 1] % np.uint64(height))
        rw = 8 + int(r[5 * k + 2] % np.uint64(max(width // 4, 9)))
        rh = 8 + int(r[5 * k + 3] % np.uint64(max(height // 4, 9)))
        img[y0:min(y0 + rh, height), x0:min(x0 + rw, width)] = near + int(r[5 * k + 4] % np.uint64(max(far - near, 1)))
    u = splitmix64(seed, width * height, start=5 * n_rects)
    img += ((u % np.uint64(5)).astype(np.int32) - 2).reshape(height, width)
    img = np.clip(img, 1, 65535).astype(np.uint16)
    sel = ((u >> np.uint64(8)) % np.uint64(10000)).reshape(height, width)
    img[sel < np.uint64(int(holes * 5000))] = 0
    starts = np.argwhere((sel >= np.uint64(5000)) & (sel < np.uint64(5000 + int(holes * 5000 / 6))))
    run = ((u >> np.uint64(24)) % np.uint64(12)).reshape(height, width)
    for y, x in starts:
        img[y, x:min(x + 1 + int(run[y, x]), width)] = 0
    return img


def depth_frames(width, height, count, first_index=0, **kw):
    return np.stack([depth_frame(width, height, first_index + i, **kw) for i in range(count)])


def rig(kind, dw, dh, ow=None, oh=None):
    """(depth intrinsics, other intrinsics, extrinsics, depth_scale) as plain tuples
    (width, height, ppx, ppy, fx, fy, model, coeffs[5]) / (rotation[9] column-major, translation[3]):
      'identity'  the same pinhole twice, no motion;
      'd435'      a D435-like pair: depth 87 deg wide (Brown-Conrady with zero coefficients = no distortion is
                  applied, model 4), colour 69 deg (inverse Brown-Conrady, model 2, zero coefficients: never used when
                  projecting), 15 mm baseline and a fraction of a degree of rotation;
      'distorted' inverse Brown-Conrady on the depth camera (deprojection polynomial, cuda-align.cu:69-77) and
                  modified Brown-Conrady on the other (projection polynomial, :31-42), 50 mm baseline, 2 degrees;
      'wild'      30-degree roll, 0.6 m offset: near rectangles leave the frame, a tile's rectangles scatter over
                  hundreds of pixels (no LDS window holds them), many are one pixel wide."""
    ow, oh = ow or dw, oh or dh
    z5 = (0.0,) * 5
    if kind == "identity":
        k = (dw, dh, dw * 0.5, dh * 0.5, dw * 0.6, dw * 0.6, 0, z5)
        return k, (ow, oh) + k[2:], ((1, 0, 0, 0, 1, 0, 0, 0, 1), (0, 0, 0)), 0.001
    if kind == "d435":
        d = (dw, dh, dw * 0.5 - 2.3, dh * 0.5 + 1.7, dw * 0.4976, dw * 0.4976, 4, z5)
        o = (ow, oh, ow * 0.5 + 3.1, oh * 0.5 - 2.2, ow * 0.7266, ow * 0.7261, 2, z5)
        a, b, c = 0.0031, -0.0042, 0.0017  # small rotation about x, y, z (first order is enough for a test rig)
        rot = (1.0, c, -b, -c, 1.0, a, b, -a, 1.0)
        return d, o, (rot, (0.0148, 0.0003, -0.0004)), 0.001
    if kind == "distorted":
        d = (dw, dh, dw * 0.5 + 1.2, dh * 0.5 - 0.8, dw * 0.62, dw * 0.618, 2, (0.08, -0.12, 0.0009, -0.0006, 0.03))
        o = (ow, oh, ow * 0.5 - 2.5, oh * 0.5 + 2.0, ow * 0.68, ow * 0.681, 1, (-0.05, 0.07, -0.0008, 0.0005, -0.01))
        cs, sn = float(np.cos(np.radians(2.0))), float(np.sin(np.radians(2.0)))
        rot = (cs, 0.0, -sn, 0.0, 1.0, 0.0, sn, 0.0, cs)  # 2 degrees about y
        return d, o, (rot, (0.05, -0.004, 0.002)), 0.001
    if kind == "wild":
        d = (dw, dh, dw * 0.5, dh * 0.5, dw * 0.5, dw * 0.5, 0, z5)
        o = (ow, oh, ow * 0.5, oh * 0.5, ow * 0.9, ow * 0.9, 0, z5)
        cs, sn = float(np.cos(np.radians(30.0))), float(np.sin(np.radians(30.0)))
        rot = (cs, sn, 0.0, -sn, cs, 0.0, 0.0, 0.0, 1.0)  # 30 degrees about z
        return d, o, (rot, (0.6, -0.2, 0.1)), 0.001
    raise ValueError(kind)
