"""CPU tests of the oracle's align_depth_to_other (cuda-align.cu:121-188, :224-280, :366-399).

The reference holds no test or fixture for it and is CUDA-only: PARITY UNPINNED by the reference.  What pins the
restatement here: (i) a case whose arithmetic is exact, worked by hand from the source; (ii) the same definition
written with numpy float32 array arithmetic (IEEE, unfused) sharing no code with the C restatement; (iii) the launch
grid quirk and the unsupported models; (iv) the committed digest (tests/golden).  No GPU."""
import ctypes as C

import numpy as np
import pytest

from orbfe import synth


def intr(mod, t):
    return mod.Intrinsics(t[0], t[1], t[2], t[3], t[4], t[5], t[6], (C.c_float * 5)(*t[7]))


def extr(mod, t):
    return mod.Extrinsics((C.c_float * 9)(*t[0]), (C.c_float * 3)(*t[1]))


def np_align(depth, scale, image_w, image_h, d, o, e):
    """The definition with numpy float32 arrays: every operation below is one IEEE float32 operation per element, in
    the order the reference writes them."""
    f = np.float32
    dw, dh, ow, oh = d[0], d[1], o[0], o[1]
    gx, gy = 32 * ((image_w + 31) // 32), 32 * ((image_h + 31) // 32)
    mx, my, rx, ry = min(gx, dw), min(gy, dh), min(gx, ow), min(gy, oh)
    rot, tr = [f(v) for v in e[0]], [f(v) for v in e[1]]
    yy, xx = np.mgrid[0:my, 0:mx]
    raw = depth[:my, :mx]
    dv = raw.astype(np.int32).astype(f) * f(scale)

    def corner(shift):
        with np.errstate(all="ignore"):
            x = ((xx.astype(f) + f(shift)) - f(d[2])) / f(d[4])
            y = ((yy.astype(f) + f(shift)) - f(d[3])) / f(d[5])
            if d[6] == 2:
                c = [f(v) for v in d[7]]
                r2 = x * x + y * y
                ff = f(1) + c[0] * r2
                ff = ff + c[1] * r2 * r2
                ff = ff + c[4] * r2 * r2 * r2
                ux = x * ff + f(2) * c[2] * x * y
                ux = ux + c[3] * (r2 + f(2) * x * x)
                uy = y * ff + f(2) * c[3] * x * y
                uy = uy + c[2] * (r2 + f(2) * y * y)
                x, y = ux, uy
            p = [dv * x, dv * y, dv]
            q = []
            for i in range(3):
                t = rot[i] * p[0] + rot[3 + i] * p[1]
                t = t + rot[6 + i] * p[2]
                t = t + tr[i]
                q.append(t)
            x = q[0] / q[2]
            y = q[1] / q[2]
            if o[6] == 1:
                c = [f(v) for v in o[7]]
                r2 = x * x + y * y
                ff = f(1) + c[0] * r2
                ff = ff + c[1] * r2 * r2
                ff = ff + c[4] * r2 * r2 * r2
                x = x * ff
                y = y * ff
                dx = x + f(2) * c[2] * x * y
                dx = dx + c[3] * (r2 + f(2) * x * x)
                dy = y + f(2) * c[3] * x * y
                dy = dy + c[2] * (r2 + f(2) * y * y)
                x, y = dx, dy
            px = (x * f(o[4]) + f(o[2])) + f(0.5)
            py = (y * f(o[5]) + f(o[3])) + f(0.5)

        def rz(v):  # cvt.rzi.s32.f32: truncate, saturate, NaN -> 0
            v = np.where(np.isnan(v), f(0), v)
            v = np.clip(v.astype(np.float64), -2147483648.0, 2147483647.0)
            return np.trunc(v).astype(np.int64)
        mxp, myp = rz(px), rz(py)
        mxp[dv == 0] = -1
        myp[dv == 0] = -1
        return mxp, myp

    ax, ay = corner(-0.5)
    bx, by = corner(0.5)
    out = np.full((oh, ow), 0xDEADBEEF, np.uint32)
    out[:ry, :rx] = 9999999
    ok = ~((ax < 0) | (ay < 0) | (bx >= ow) | (by >= oh))
    for y, x in np.argwhere(ok):
        if ax[y, x] <= bx[y, x] and ay[y, x] <= by[y, x]:
            blk = out[ay[y, x]:by[y, x] + 1, ax[y, x]:bx[y, x] + 1]
            np.minimum(blk, raw[y, x], out=blk)
    reg = out[:ry, :rx]
    reg[reg == 9999999] = 0
    return out, np.stack([np.stack([ax, ay], -1), np.stack([bx, by], -1)])


def random_case(rng, trial):
    """One seeded random (depth intrinsics, other intrinsics, extrinsics, scale, frames): focal lengths 0.3 .. 1.2 x the
    width on both sides (rectangles from 1 x 1 to 5 x 5 pixels), principal points off centre, small random rotations about
    all three axes, translations up to 0.3 m, random distortion models and coefficients, hole rates, depth ranges, sizes."""
    dw, dh = int(rng.integers(8, 300)), int(rng.integers(8, 200))
    ow, oh = (dw, dh) if rng.random() < 0.4 else (int(rng.integers(8, 400)), int(rng.integers(8, 260)))
    n = int(rng.integers(1, 12))
    co = lambda: tuple(float(v) for v in rng.normal(0, [0.06, 0.08, 0.001, 0.001, 0.02]))
    d = (dw, dh, dw * rng.uniform(0.4, 0.6), dh * rng.uniform(0.4, 0.6), dw * rng.uniform(0.3, 1.2), dw * rng.uniform(0.3, 1.2),
         int(rng.choice([0, 2, 4])), co())
    o = (ow, oh, ow * rng.uniform(0.4, 0.6), oh * rng.uniform(0.4, 0.6), ow * rng.uniform(0.3, 1.2), ow * rng.uniform(0.3, 1.2),
         int(rng.choice([0, 1, 2, 4])), co())
    a, b, c = rng.normal(0, 0.03, 3)
    rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    rot = (rz @ ry @ rx).astype(np.float32)
    e = (tuple(float(v) for v in rot.T.reshape(-1)), tuple(float(v) for v in rng.uniform(-0.3, 0.3, 3)))  # column-major
    scale = float(rng.choice([0.001, 0.0001, 0.00025]))
    frames = synth.depth_frames(dw, dh, n, first_index=1000 + 13 * trial, n_rects=int(rng.integers(0, 30)),
                                holes=float(rng.uniform(0, 0.6)), near=int(rng.integers(1, 2000)), far=int(rng.integers(2000, 60000)))
    return d, o, e, scale, frames


def test_fuzz_against_the_numpy_definition(oracle_mod):
    """24 seeded random rigs (random_case): map and image of the C restatement equal the numpy float32 definition."""
    rng = np.random.default_rng(20261005)
    covered = []
    for trial in range(24):
        d, o, e, scale, frames = random_case(rng, trial)
        iw, ih = max(d[0], o[0]), max(d[1], o[1])
        out, pm = oracle_mod.align_depth_to_other(frames[0], scale, iw, ih, intr(oracle_mod, d), intr(oracle_mod, o),
                                                  extr(oracle_mod, e), want_map=True)
        want, wmap = np_align(frames[0], scale, iw, ih, d, o, e)
        np.testing.assert_array_equal(pm, wmap.astype(np.int32), err_msg="trial %d" % trial)
        np.testing.assert_array_equal(out, want, err_msg="trial %d" % trial)
        covered.append((out != 0).mean())
    assert max(covered) > 0.8 and min(covered) < 0.2  # dense and sparse outcomes both occur


def test_exact_case_worked_by_hand(oracle_mod):
    """fx = fy = 1, pp = 0, identity motion, depth_val a power of two: every float operation is exact, so corner
    -0.5 of pixel (x, y) maps to int((x - 0.5) + 0.5) = x and corner +0.5 to x + 1 (cuda-align.cu:143-155): the pixel's
    depth goes to the 2 x 2 block [x, x + 1] x [y, y + 1], unless x + 1 or y + 1 leaves the image (:241) -- then nowhere."""
    w, h = 9, 7
    rng = np.random.default_rng(3)
    depth = (1 << rng.integers(4, 12, (h, w))).astype(np.uint16)  # raw * 2^-10 is a power of two
    depth[rng.random((h, w)) < 0.3] = 0
    k = (w, h, 0.0, 0.0, 1.0, 1.0, 0, (0,) * 5)
    e = ((1, 0, 0, 0, 1, 0, 0, 0, 1), (0, 0, 0))
    out, pm = oracle_mod.align_depth_to_other(depth, 2.0 ** -10, w, h, intr(oracle_mod, k), intr(oracle_mod, k),
                                              extr(oracle_mod, e), want_map=True)
    want = np.zeros((h, w), np.uint32)
    for v in range(h):
        for u in range(w):
            best = 0
            for y in (v - 1, v):
                for x in (u - 1, u):
                    if 0 <= x <= w - 2 and 0 <= y <= h - 2 and depth[y, x]:
                        best = depth[y, x] if best == 0 else min(best, depth[y, x])
            want[v, u] = best
    np.testing.assert_array_equal(out, want)
    yy, xx = np.mgrid[0:h, 0:w]
    valid = depth != 0
    assert (pm[0][valid] == np.stack([xx, yy], -1)[valid]).all()
    assert (pm[1][valid] == np.stack([xx + 1, yy + 1], -1)[valid]).all()
    assert (pm[0][~valid] == -1).all() and (pm[1][~valid] == -1).all()  # :138-140


@pytest.mark.parametrize("kind", ["identity", "d435", "distorted", "wild"])
@pytest.mark.parametrize("size", [(96, 64, 96, 64), (101, 67, 80, 60), (64, 48, 131, 77)])
def test_against_the_numpy_definition(oracle_mod, kind, size):
    dw, dh, ow, oh = size
    d, o, e, scale = synth.rig(kind, dw, dh, ow, oh)
    depth = synth.depth_frame(dw, dh, index=dw + len(kind), n_rects=12)
    iw, ih = max(dw, ow), max(dh, oh)
    out, pm = oracle_mod.align_depth_to_other(depth, scale, iw, ih, intr(oracle_mod, d), intr(oracle_mod, o),
                                              extr(oracle_mod, e), want_map=True)
    want, wmap = np_align(depth, scale, iw, ih, d, o, e)
    np.testing.assert_array_equal(pm, wmap.astype(np.int32))
    np.testing.assert_array_equal(out, want)
    if kind != "wild":
        assert (out != 0).mean() > 0.3


def test_zero_z_gives_saturated_and_nan_pixels(oracle_mod):
    """other_point z = 0 (translation z = -depth): x / 0 = +-inf, 0 / 0 = NaN.  CUDA's float -> int conversion
    saturates and maps NaN to 0 (PTX cvt.rzi.s32.f32), so such a pixel is dropped by the bounds test unless both of
    its corners are NaN -> (0, 0): then it lands on output pixel (0, 0)."""
    w, h = 16, 8
    depth = np.full((h, w), 1024, np.uint16)
    depth[3, 5] = 512
    k = (w, h, 0.0, 0.0, 1.0, 1.0, 0, (0,) * 5)
    e = ((1, 0, 0, 0, 1, 0, 0, 0, 1), (0, 0, -1.0))  # z' = 1.0 - 1.0 = 0 for raw 1024 at scale 2^-10
    out, pm = oracle_mod.align_depth_to_other(depth, 2.0 ** -10, w, h, intr(oracle_mod, k), intr(oracle_mod, k),
                                              extr(oracle_mod, e), want_map=True)
    want, wmap = np_align(depth, 2.0 ** -10, w, h, k, k, e)
    np.testing.assert_array_equal(pm, wmap.astype(np.int32))
    np.testing.assert_array_equal(out, want)
    # pixel (0, 0): corner -0.5 -> (-0.5 / 0, -0.5 / 0) = -inf -> INT_MIN < 0: dropped
    assert tuple(pm[0][0, 0]) == (-2147483648, -2147483648)
    # pixel (1, 1): corner +0.5 -> +inf -> INT_MAX >= width: dropped
    assert tuple(pm[1][1, 1]) == (2147483647, 2147483647)


def test_the_launch_grid_bounds_what_is_read_and_reset(oracle_mod):
    """The four launches share one grid made from image_width / image_height (cuda-align.cu:378-380), every bound
    inside the kernels is an intrinsics field: with image 40 x 20 (grid 64 x 32) on 100 x 50 images, depth pixels with
    x >= 64 or y >= 32 are never mapped and output pixels there are neither reset nor zeroed -- but splats reach them."""
    dw, dh = 100, 50
    d, o, e, scale = synth.rig("d435", dw, dh)
    depth = synth.depth_frame(dw, dh, 5, n_rects=6, holes=0.0)
    before = np.full((dh, dw), 77777, np.uint32)
    before[::2] = 3  # smaller than any depth: stays where the grid does not reset
    out, pm = oracle_mod.align_depth_to_other(depth, scale, 40, 20, intr(oracle_mod, d), intr(oracle_mod, o),
                                              extr(oracle_mod, e), out_init=before, want_map=True)
    want, _ = np_align(depth, scale, 40, 20, d, o, e)
    assert (pm[:, 32:] == -7).all() and (pm[:, :, 64:] == -7).all()  # never written
    mask = want == 0xDEADBEEF  # np_align's marker for "not reset, no splat"
    assert mask[:32, :64].sum() == 0 and mask.sum() > 0
    # outside the grid: min(before, splats)
    outside = np.ones((dh, dw), bool)
    outside[:32, :64] = False
    splat = np.where(mask, np.uint32(0xFFFFFFFF), want)
    np.testing.assert_array_equal(out[outside], np.minimum(before, splat)[outside])
    np.testing.assert_array_equal(out[:32, :64], want[:32, :64])
    assert (out[outside] < 77777).any() and (out[outside] == 77777).any()


def test_ftheta_projection_against_libm(oracle_mod):
    """Other-camera model 3 (RS2_DISTORTION_FTHETA, cuda-align.cu:44-50): the oracle evaluates the branch in float with
    the build's deterministic atanf / tanf (include/orbfe_math.h: the float overloads are what nvcc picks for float
    operands).  Against the same formula with numpy's float64 atan / tan: the mapped pixel may differ only where the
    exact value sits within rounding distance of a half-integer, i.e. almost nowhere; the output images agree on
    > 99.5 % of the pixels and never by more than one depth pixel's neighbourhood."""
    dw, dh = 160, 120
    d, o, e, scale = synth.rig("d435", dw, dh)
    o = list(o)
    o[6] = 3
    o[7] = [0.92, 0.0, 0.0, 0.0, 0.0]  # coeffs[0] = the lens' field-of-view parameter (radians)
    depth = synth.depth_frame(dw, dh, 11)
    got, pmap = oracle_mod.align_depth_to_other(depth, scale, dw, dh, intr(oracle_mod, d), intr(oracle_mod, o), extr(oracle_mod, e),
                                                want_map=True)
    # the same with float64 transcendental functions
    f = np.float32
    yy, xx = np.mgrid[0:dh, 0:dw]
    dv = depth.astype(np.int32).astype(f) * f(scale)
    rot, tr = [f(v) for v in e[0]], [f(v) for v in e[1]]
    out = {}
    with np.errstate(all="ignore"):
        for z, shift in ((0, -0.5), (1, 0.5)):
            x = ((xx.astype(f) + f(shift)) - f(d[2])) / f(d[4])
            y = ((yy.astype(f) + f(shift)) - f(d[3])) / f(d[5])
            p = [dv * x, dv * y, dv]
            q = [rot[i] * p[0] + rot[3 + i] * p[1] + rot[6 + i] * p[2] + tr[i] for i in range(3)]
            x, y = (q[0] / q[2]).astype(np.float64), (q[1] / q[2]).astype(np.float64)
            r = np.sqrt(x * x + y * y)
            rd = 1.0 / o[7][0] * np.arctan(2 * r * np.tan(o[7][0] / 2.0))
            x, y = x * (rd / r), y * (rd / r)
            out[z] = (np.trunc(x * o[4] + o[2] + 0.5), np.trunc(y * o[5] + o[3] + 0.5))
    m = pmap
    valid = depth != 0
    agree = 0
    for z in (0, 1):
        agree += ((m[z, :, :, 0] == out[z][0]) & (m[z, :, :, 1] == out[z][1]))[valid].sum()
    assert agree >= 0.995 * 2 * valid.sum(), (agree, valid.sum())
    assert (got != 0).mean() > 0.3  # the rig still lands inside the image


def test_models_the_reference_cannot_run(oracle_mod):
    d, o, e, scale = synth.rig("identity", 32, 32)
    depth = np.ones((32, 32), np.uint16)
    for dm, om in [(1, 0), (3, 0), (1, 3)]:
        dd, oo = list(d), list(o)
        dd[6], oo[6] = dm, om
        with pytest.raises(ValueError):
            oracle_mod.align_depth_to_other(depth, scale, 32, 32, intr(oracle_mod, dd), intr(oracle_mod, oo),
                                            extr(oracle_mod, e))
