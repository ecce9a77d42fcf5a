"""ctypes binding of the CPU oracle (oracle/orbfe_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Parity is unpinned by the reference (it holds no tests or fixtures for this path); see
orbfe_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liborbfe_oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("orbfe_oracle.c", "orbfe_oracle.h")]
    src += [os.path.join(_HERE, "..", "include", f) for f in ("orbfe_math.h", "orbfe_pattern.h")]
    stale = force or not os.path.exists(_LIB) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src)
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB


class Level(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("image_pitch", C.c_int),
                ("image", C.c_void_p), ("response_pitch", C.c_int), ("response", C.c_void_p)]


class Config(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("levels", C.c_int), ("cell", C.c_int),
                ("fast_threshold", C.c_float), ("min_arc", C.c_int), ("max_features", C.c_int),
                ("angle_in_radians", C.c_int), ("descriptor_level", C.c_int)]


KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("score", "<f4"), ("level", "<i4"),
                           ("angle", "<f4"), ("desc", "u1", (32,))])
assert KEYPOINT_DTYPE.itemsize == 52

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.oracle_atan2f.restype = C.c_float
        _lib.oracle_atan2f.argtypes = [C.c_float, C.c_float]
        _lib.oracle_sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        _lib.oracle_pattern.restype = C.POINTER(C.c_int8)
        _lib.oracle_level_dims.restype = C.c_size_t
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def make_config(width, height, levels=1, cell=32, fast_threshold=13.0, min_arc=12,
                max_features=0, angle_in_radians=0, descriptor_level=0):
    return Config(width, height, levels, cell, fast_threshold, min_arc, max_features,
                  angle_in_radians, descriptor_level)


def level_dims(width, height, level):
    return width >> level, height >> level


def num_cells(width, height, cell=32):
    return ((width + cell - 1) // cell) * ((height + cell - 1) // cell)


def pattern():
    p = lib().oracle_pattern()
    return np.ctypeslib.as_array(p, shape=(1024,)).copy()


def atan2f(y, x):
    return np.float32(lib().oracle_atan2f(np.float32(y), np.float32(x)))


def sincosf(x):
    s, c = C.c_float(), C.c_float()
    lib().oracle_sincosf(np.float32(x), C.byref(s), C.byref(c))
    return np.float32(s.value), np.float32(c.value)


def has_arc(mask, arc):
    return int(lib().oracle_has_arc(C.c_uint32(mask), arc))


def fast_is_corner(mask, arc):
    return int(lib().oracle_fast_is_corner(C.c_uint32(mask), arc))


def rgb_to_grayscale(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    out = np.zeros((h, w), np.uint8)
    lib().oracle_rgb_to_grayscale(_p(out), _p(rgb), w, h, w, 3 * w)
    return out


def gaussian_blur_3x3(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.full((h, w), 0xCD, dtype=np.uint8)
    lib().oracle_gaussian_blur_3x3(_p(out), w, _p(img), w, w, h)
    return out


def halfsample(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.zeros((h // 2, w // 2), dtype=np.uint8)
    lib().oracle_halfsample(_p(img), w, _p(out), max(w // 2, 1), w // 2, h // 2)
    return out


def fast_lut(min_arc):
    lut = np.zeros(65536, dtype=np.uint8)
    lib().oracle_fast_calculate_lut(_p(lut), min_arc)
    return lut


def fast_response(img, lut, threshold=13.0, border=3, score=1):
    """score: the reference's enum fast_score (0 SUM_OF_ABS_DIFF_ALL, 1 SUM_OF_ABS_DIFF_ON_ARC, 2 MAX_THRESHOLD)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    resp = np.full((h, w), -1.0, dtype=np.float32)
    lib().oracle_fast_calc_corner_response_score(w, h, w, _p(img), border, border, _p(lut),
                                                 C.c_float(threshold), score, w, _p(resp))
    return resp


def _levels_struct(images, responses):
    arr = (Level * len(images))()
    for i, (im, rs) in enumerate(zip(images, responses)):
        h, w = im.shape
        arr[i] = Level(w, h, max(w, 1), im.ctypes.data, max(w, 1),
                       rs.ctypes.data if rs is not None else None)
    return arr


def grid_nms(responses, cell=32):
    """responses: list of f32 [H_l, W_l] arrays, level 0 first."""
    responses = [np.ascontiguousarray(r, dtype=np.float32) for r in responses]
    images = [np.zeros(r.shape, dtype=np.uint8) for r in responses]
    h, w = responses[0].shape
    k = num_cells(w, h, cell)
    pos = np.full((k, 2), -7.0, dtype=np.float32)
    score = np.full(k, -7.0, dtype=np.float32)
    level = np.full(k, -7, dtype=np.int32)
    lv = _levels_struct(images, responses)
    lib().oracle_grid_nms(lv, len(responses), cell, _p(pos), _p(score), _p(level))
    return pos, score, level


def compute_fast_angle(pos, score, img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    h, w = img.shape
    n = pos.shape[0]
    angle = np.zeros(n, dtype=np.float32)
    sc = np.ascontiguousarray(score, dtype=np.float32) if score is not None else None
    lib().oracle_compute_fast_angle(_p(angle), _p(pos), _p(sc), _p(img), w, w, h, n)
    return angle


def calc_orb(angle, pos, img, angle_in_radians=0, fma=False):
    """fma=True: GET_VALUE's sums contracted into FMAs (exposure probe, see oracle_calc_orb_fma)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    angle = np.ascontiguousarray(angle, dtype=np.float32)
    h, w = img.shape
    n = pos.shape[0]
    desc = np.zeros((n, 32), dtype=np.uint8)
    d32 = np.zeros(n, dtype=np.uint32)
    fn = lib().oracle_calc_orb_fma if fma else lib().oracle_calc_orb
    fn(_p(angle), _p(pos), _p(desc), _p(d32), _p(img), w, w, h, n, angle_in_radians)
    return desc, d32


class Intrinsics(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("ppx", C.c_float), ("ppy", C.c_float),
                ("fx", C.c_float), ("fy", C.c_float), ("model", C.c_int32), ("coeffs", C.c_float * 5)]


def keypoint_pixel_to_point(depth, intrin, pos, score, desc, fix_depth_index=0):
    depth = np.ascontiguousarray(depth, dtype=np.uint32)
    pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 2)
    score = np.ascontiguousarray(score, dtype=np.float32)
    desc = np.ascontiguousarray(desc, dtype=np.uint32)
    h, w = depth.shape
    n = pos.shape[0]
    pos_out = np.zeros((max(n, 1), 2), np.float32)
    pts = np.zeros((max(n, 1), 3), np.float64)
    dout = np.zeros(max(n, 1), np.uint32)
    cnt = lib().oracle_keypoint_pixel_to_point(_p(depth), C.byref(intrin), w, h, _p(pos_out), _p(pos), _p(score),
                                               _p(pts), _p(dout), _p(desc), n, fix_depth_index)
    return pos_out[:cnt], pts[:cnt], dout[:cnt], cnt


class Extrinsics(C.Structure):
    _fields_ = [("rotation", C.c_float * 9), ("translation", C.c_float * 3)]


def align_depth_to_other(depth, depth_scale, image_width, image_height, depth_intrin, other_intrin, extrin,
                         out_init=None, want_map=False):
    """depth u16[dh, dw] -> aligned u32[oh, ow] (cuda-align.cu:366-399).  out_init: what the caller's output
    buffer held before the call (matters only outside the launch grid); default 0xDEADBEEF.
    Returns (aligned, map int32[2, dh, dw, 2] | None); raises on the models the reference cannot run."""
    depth = np.ascontiguousarray(depth, dtype=np.uint16)
    assert depth.shape == (depth_intrin.height, depth_intrin.width)
    out = np.full((other_intrin.height, other_intrin.width), 0xDEADBEEF, np.uint32) if out_init is None \
        else np.ascontiguousarray(out_init, dtype=np.uint32).copy()
    pm = np.full((2,) + depth.shape + (2,), -7, np.int32) if want_map else None
    rc = lib().oracle_align_depth_to_other(_p(out), _p(depth), _p(pm), C.c_float(depth_scale), image_width,
                                           image_height, C.byref(depth_intrin), C.byref(other_intrin),
                                           C.byref(extrin))
    if rc != 0:
        raise ValueError("align_depth_to_other: unsupported distortion model")
    return out, pm


def match_keypoints(pos_prev, desc_prev, pos_curr, desc_curr, max_px=2, max_ham=4):
    pos_prev = np.ascontiguousarray(pos_prev, dtype=np.float32).reshape(-1, 2)
    pos_curr = np.ascontiguousarray(pos_curr, dtype=np.float32).reshape(-1, 2)
    desc_prev = np.ascontiguousarray(desc_prev, dtype=np.uint32)
    desc_curr = np.ascontiguousarray(desc_curr, dtype=np.uint32)
    n_prev, n_curr = pos_prev.shape[0], pos_curr.shape[0]
    idx = np.full(max(n_prev, 1), -9, dtype=np.int32)
    n = lib().oracle_match_keypoints(_p(pos_prev), _p(desc_prev), n_prev, _p(pos_curr),
                                     _p(desc_curr), n_curr, max_px, max_ham, _p(idx))
    return idx[:n_prev], n


def match_compact(match_idx, pos_curr, points_prev=None, points_curr=None):
    """-> (keypoints_x u16[n], keypoints_y u16[n], prev_matched f64[n,3] | None, curr_matched | None)"""
    match_idx = np.ascontiguousarray(match_idx, dtype=np.int32)
    pos_curr = np.ascontiguousarray(pos_curr, dtype=np.float32).reshape(-1, 2)
    n_prev = len(match_idx)
    kx = np.zeros(max(n_prev, 1), np.uint16)
    ky = np.zeros(max(n_prev, 1), np.uint16)
    pm = cm = None
    if points_prev is not None:
        points_prev = np.ascontiguousarray(points_prev, dtype=np.float64).reshape(-1, 3)
        points_curr = np.ascontiguousarray(points_curr, dtype=np.float64).reshape(-1, 3)
        pm = np.zeros((max(n_prev, 1), 3))
        cm = np.zeros((max(n_prev, 1), 3))
    n = lib().oracle_match_compact(_p(match_idx), n_prev, _p(points_prev), _p(points_curr), _p(pos_curr), _p(pm), _p(cm),
                                   _p(kx), _p(ky))
    return kx[:n], ky[:n], (pm[:n] if pm is not None else None), (cm[:n] if cm is not None else None)


def reproject_points(points_prev, T, intrin):
    """points_prev f64[n,3], T 4x4 (row-major numpy, converted to Eigen's column-major), intrin = Intrinsics"""
    points_prev = np.ascontiguousarray(points_prev, dtype=np.float64).reshape(-1, 3)
    Tc = np.ascontiguousarray(np.asarray(T, dtype=np.float64).T).reshape(-1)  # column-major
    out = np.zeros((max(len(points_prev), 1), 2), np.float32)
    lib().oracle_reproject_points(_p(out), _p(points_prev), len(points_prev), _p(Tc), C.byref(intrin))
    return out[:len(points_prev)]


def match256(desc_a, desc_b, pos_a=None, pos_b=None, window=-1, max_dist=256):
    desc_a = np.ascontiguousarray(desc_a, dtype=np.uint8).reshape(-1, 32)
    desc_b = np.ascontiguousarray(desc_b, dtype=np.uint8).reshape(-1, 32)
    na, nb = desc_a.shape[0], desc_b.shape[0]
    if pos_a is not None:
        pos_a = np.ascontiguousarray(pos_a, dtype=np.float32).reshape(-1, 2)
        pos_b = np.ascontiguousarray(pos_b, dtype=np.float32).reshape(-1, 2)
    idx = np.full(max(na, 1), -9, dtype=np.int32)
    dist = np.full(max(na, 1), -9, dtype=np.int32)
    lib().oracle_match256(_p(desc_a), _p(pos_a), na, _p(desc_b), _p(pos_b), nb, window,
                          max_dist, _p(idx), _p(dist))
    return idx[:na], dist[:na]


def extract_frame(gray, cfg, want_pyramid=False):
    """Run the whole pipeline on one frame.  Returns a dict of numpy arrays."""
    gray = np.ascontiguousarray(gray, dtype=np.uint8)
    h, w = gray.shape
    assert (w, h) == (cfg.width, cfg.height)
    k = num_cells(w, h, cfg.cell)
    out = {
        "pos": np.zeros((k, 2), np.float32), "score": np.zeros(k, np.float32),
        "level": np.zeros(k, np.int32), "angle": np.zeros(k, np.float32),
        "desc": np.zeros((k, 32), np.uint8), "desc32": np.zeros(k, np.uint32),
    }
    records = np.zeros(k, dtype=KEYPOINT_DTYPE)
    pyr = None
    pyr_ptrs = None
    if want_pyramid:
        pyr = [np.zeros((h >> l, w >> l), np.uint8) for l in range(cfg.levels)]
        pyr_ptrs = (C.c_void_p * cfg.levels)(*[p.ctypes.data if p.size else None for p in pyr])
    n = lib().oracle_extract_frame(C.byref(cfg), _p(gray), w, pyr_ptrs, _p(out["pos"]),
                                   _p(out["score"]), _p(out["level"]), _p(out["angle"]),
                                   _p(out["desc"]), _p(out["desc32"]), _p(records))
    out["records"] = records[:n].copy()
    out["count"] = n
    if want_pyramid:
        out["pyramid"] = pyr
    return out
