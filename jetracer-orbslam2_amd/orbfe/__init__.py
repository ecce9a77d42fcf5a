"""orbfe -- Python binding (ctypes) of liborbfe.so, the MI355X-native ORB front end.

The product is the HIP library behind the C ABI in include/orbfe.h; this module only loads
it and declares the signatures, for the tests and bench.  There is no CPU fallback: if the
library is missing, `lib()` raises.  Function names mirror the reference's free functions
(src/cuda/*.cuh of dsvua/jetracer-orbslam2): gaussian_blur_3x3, pyramid_create_levels,
fast_gpu_calculate_lut -> fast_calculate_lut, fast_gpu_calc_corner_response ->
fast_calc_corner_response, grid_nms, detect, compute_fast_angle, calc_orb, match_keypoints.
All pointer arguments are raw device addresses (ints), e.g. torch.Tensor.data_ptr().
"""
import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# ORBFE_LIB: another build of the same library (A/B timing of kernel variants); still a HIP build, never a fallback
LIB_PATH = os.environ.get("ORBFE_LIB") or os.path.join(PKG_DIR, "liborbfe.so")

OK, ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_NO_DEVICE, ERR_CAPACITY = range(6)
SUM_OF_ABS_DIFF_ALL, SUM_OF_ABS_DIFF_ON_ARC, MAX_THRESHOLD = 0, 1, 2

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("score", "<f4"), ("level", "<i4"),
                           ("angle", "<f4"), ("desc", "u1", (32,))])
assert KEYPOINT_DTYPE.itemsize == 52


class OrbfeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("orbfe error %d: %s" % (code, msg))
        self.code = code


class PyramidLevel(C.Structure):  # orbfe_pyramid_level == reference pyramid_t
    _fields_ = [("image_width", C.c_size_t), ("image_height", C.c_size_t),
                ("image_pitch", C.c_size_t), ("image", C.c_void_p),
                ("response_pitch", C.c_size_t), ("response", C.c_void_p)]


class Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("levels", C.c_int32),
                ("cell", C.c_int32), ("fast_threshold", C.c_int32), ("min_arc", C.c_int32),
                ("max_features", C.c_int32), ("angle_in_radians", C.c_int32),
                ("max_batch", C.c_int32), ("device", C.c_int32), ("descriptor_level", C.c_int32)]


class Intrinsics(C.Structure):  # orbfe_intrinsics == rs2_intrinsics
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("ppx", C.c_float), ("ppy", C.c_float),
                ("fx", C.c_float), ("fy", C.c_float), ("model", C.c_int32), ("coeffs", C.c_float * 5)]


class Extrinsics(C.Structure):  # orbfe_extrinsics == rs2_extrinsics
    _fields_ = [("rotation", C.c_float * 9), ("translation", C.c_float * 3)]


class Soa(C.Structure):
    _fields_ = [("d_pos", C.c_void_p), ("d_score", C.c_void_p), ("d_level", C.c_void_p),
                ("d_angle", C.c_void_p), ("d_desc", C.c_void_p), ("d_desc32", C.c_void_p)]


_SIGS = {
    "orbfe_version": (C.c_int, []),
    "orbfe_device_count": (C.c_int, []),
    "orbfe_load_pattern": (C.c_int, []),
    "orbfe_rgb_to_grayscale": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "orbfe_gaussian_blur_3x3": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                          C.c_int, C.c_void_p]),
    "orbfe_pyramid_create_levels": (C.c_int, [C.POINTER(PyramidLevel), C.c_int, C.c_void_p]),
    "orbfe_fast_calculate_lut": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "orbfe_fast_calc_corner_response": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                                  C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_int,
                                                  C.c_int, C.c_void_p, C.c_void_p]),
    "orbfe_grid_nms": (C.c_int, [C.POINTER(PyramidLevel), C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    "orbfe_detect": (C.c_int, [C.POINTER(PyramidLevel), C.c_int, C.c_void_p, C.c_float,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orbfe_compute_fast_angle": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_void_p]),
    "orbfe_calc_orb": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "orbfe_match_keypoints": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "orbfe_keypoint_pixel_to_point": (C.c_int, [C.c_void_p, C.POINTER(Intrinsics), C.c_int, C.c_int, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "orbfe_align_depth_to_other": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int,
                                             C.POINTER(Intrinsics), C.POINTER(Intrinsics), C.POINTER(Extrinsics),
                                             C.c_void_p]),
    "orbfe_align_depth_batch": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_float,
                                          C.POINTER(Intrinsics), C.POINTER(Intrinsics), C.POINTER(Extrinsics),
                                          C.c_void_p]),
    "orbfe_match_compact": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orbfe_reproject_points": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(Intrinsics),
                                         C.c_void_p]),
    "orbfe_match256": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                 C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orbfe_default_config": (None, [C.POINTER(Config), C.c_int, C.c_int]),
    "orbfe_create": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "orbfe_destroy": (None, [C.c_void_p]),
    "orbfe_last_error": (C.c_char_p, [C.c_void_p]),
    "orbfe_num_cells": (C.c_int, [C.c_void_p]),
    "orbfe_max_keypoints": (C.c_int, [C.c_void_p]),
    "orbfe_num_levels": (C.c_int, [C.c_void_p]),
    "orbfe_level_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.POINTER(C.c_size_t), C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_size_t)]),
    "orbfe_build_pyramid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int,
                                      C.c_void_p]),
    "orbfe_build_pyramid_rgb": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int,
                                          C.c_void_p]),
    "orbfe_extract_rgb": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int,
                                    C.c_void_p, C.c_void_p, C.POINTER(Soa), C.c_void_p]),
    "orbfe_detect_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "orbfe_detect_batch_shard": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "orbfe_export_cell_keys": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "orbfe_import_cell_keys": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "orbfe_describe_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                       C.POINTER(Soa), C.c_void_p]),
    "orbfe_extract": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int,
                                C.c_void_p, C.c_void_p, C.POINTER(Soa), C.c_void_p]),
    "orbfe_match_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orbfe_match_pairs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orbfe_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "orbfe_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "orbfe_stream_sync": (C.c_int, [C.c_void_p]),
    "orbfe_dispatch_info": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "orbfe_layout_bounds": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong),
                                      C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_int)]),
    "orbfe_selfcheck_steer_table": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
}

EXPORTS = tuple(_SIGS)  # every symbol include/orbfe.h declares


class FrameMessage(C.Structure):  # orbfe_frame_message, include/orbfe_wire.h
    _fields_ = [("theta", C.c_float * 3), ("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32),
                ("keypoints_x", C.c_void_p), ("keypoints_y", C.c_void_p), ("matched_keypoints", C.c_int32),
                ("image", C.c_void_p), ("image_length", C.c_size_t)]


_WIRE_SIGS = {  # include/orbfe_wire.h: the result record's wire format (host code inside liborbfe.so)
    "orbfe_bson_new": (C.c_void_p, []),
    "orbfe_bson_free": (None, [C.c_void_p]),
    "orbfe_bson_add": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_size_t]),
    "orbfe_bson_process": (C.c_int, [C.c_void_p]),
    "orbfe_bson_ptr": (C.c_void_p, [C.c_void_p]),
    "orbfe_bson_size": (C.c_uint32, [C.c_void_p]),
    "orbfe_wire_angles": (None, [C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "orbfe_wire_frame_size": (C.c_size_t, [C.POINTER(FrameMessage)]),
    "orbfe_wire_frame_encode": (C.c_int, [C.POINTER(FrameMessage), C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "orbfe_bson_find": (C.c_int, [C.c_void_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
}
WIRE_EXPORTS = tuple(_WIRE_SIGS)


class Imu(C.Structure):  # orbfe_imu, include/orbfe_pose.h
    _fields_ = [("theta", C.c_float * 3), ("alpha", C.c_float), ("last_ts_gyro", C.c_double),
                ("first_gyro", C.c_int32), ("first_accel", C.c_int32)]


_POSE_SIGS = {  # include/orbfe_pose.h: f4, host code inside liborbfe.so
    "orbfe_best_fit_transform": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    "orbfe_icp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "orbfe_imu_init": (None, [C.POINTER(Imu)]),
    "orbfe_imu_process_gyro": (None, [C.POINTER(Imu), C.POINTER(C.c_float), C.c_double]),
    "orbfe_imu_process_accel": (None, [C.POINTER(Imu), C.POINTER(C.c_float)]),
}
POSE_EXPORTS = tuple(_POSE_SIGS)


class IngestConfig(C.Structure):  # orbfe_ingest_config, include/orbfe_ingest.h
    _fields_ = [("slots", C.c_int32), ("frames_per_slot", C.c_int32), ("channels", C.c_int32), ("match_mode", C.c_int32),
                ("match_window", C.c_int32), ("match_max_distance", C.c_int32), ("download_matches", C.c_int32),
                ("reserved", C.c_int32)]


_INGEST_SIGS = {  # include/orbfe_ingest.h: the pinned host <-> device staging ring (host code inside liborbfe.so)
    "orbfe_ingest_default_config": (None, [C.POINTER(IngestConfig), C.c_int]),
    "orbfe_ingest_create": (C.c_int, [C.c_void_p, C.POINTER(IngestConfig), C.POINTER(C.c_void_p)]),
    "orbfe_ingest_destroy": (None, [C.c_void_p]),
    "orbfe_ingest_last_error": (C.c_char_p, [C.c_void_p]),
    "orbfe_ingest_slots": (C.c_int, [C.c_void_p]),
    "orbfe_ingest_frame_bytes": (C.c_size_t, [C.c_void_p]),
    "orbfe_ingest_host_frames": (C.c_void_p, [C.c_void_p, C.c_int]),
    "orbfe_ingest_submit": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "orbfe_ingest_submit_from": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t]),
    "orbfe_ingest_ready": (C.c_int, [C.c_void_p, C.c_int]),
    "orbfe_ingest_wait": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_void_p)]),
    "orbfe_ingest_device_buffers": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                              C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "orbfe_ingest_compute_stream": (C.c_void_p, [C.c_void_p]),
    "orbfe_ingest_timing": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                      C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
}
INGEST_EXPORTS = tuple(_INGEST_SIGS)

_lib = None


def source_hash():
    """sha256 over the sources liborbfe.so is built from (csrc/*.hip, *.hpp, *.cpp and include/*.h, by name): stamps
    PMC-derived numbers (profiles/traffic.json) so that bench.py can tell when the kernels have changed since."""
    import glob
    import hashlib
    root = os.path.dirname(PKG_DIR)
    files = sorted(glob.glob(os.path.join(PKG_DIR, "csrc", "*.hip")) + glob.glob(os.path.join(PKG_DIR, "csrc", "*.hpp")) +
                   glob.glob(os.path.join(PKG_DIR, "csrc", "*.cpp")) + glob.glob(os.path.join(root, "include", "*.h")))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def lib():
    """Load liborbfe.so; raise if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OrbfeError(-1, "liborbfe.so not found at %s: build it with "
                                 "`python -c 'import __graft_entry__ as g; g.build()'` or "
                                 "`make -C jetracer-orbslam2_amd/csrc` (needs hipcc; there is "
                                 "no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in list(_SIGS.items()) + list(_WIRE_SIGS.items()) + list(_POSE_SIGS.items()) + list(_INGEST_SIGS.items()):
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code, ctx=None):
    if code != OK:
        msg = lib().orbfe_last_error(ctx)
        raise OrbfeError(code, msg.decode() if msg else "")


def make_levels(images, responses=None):
    """images/responses: lists of (device_ptr, width, height, pitch_bytes)."""
    arr = (PyramidLevel * len(images))()
    for i, (ptr, w, h, pitch) in enumerate(images):
        rp, rpitch = (0, 0)
        if responses is not None:
            rp, _, _, rpitch = responses[i]
        arr[i] = PyramidLevel(w, h, pitch, ptr, rpitch, rp)
    return arr


class Context:
    """RAII wrapper of orbfe_ctx (batch API)."""

    def __init__(self, width, height, levels=1, cell=32, fast_threshold=13, min_arc=12,
                 max_features=0, angle_in_radians=0, max_batch=1, device=0, descriptor_level=0):
        L = lib()
        self.cfg = Config(width, height, levels, cell, fast_threshold, min_arc, max_features,
                          angle_in_radians, max_batch, device, descriptor_level)
        h = C.c_void_p()
        check(L.orbfe_create(C.byref(self.cfg), C.byref(h)))
        self.handle = h
        self.K = L.orbfe_num_cells(h)
        self.cap = L.orbfe_max_keypoints(h)
        self.levels = L.orbfe_num_levels(h)

    def close(self):
        if getattr(self, "handle", None):
            lib().orbfe_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def level_info(self, level):
        w, h = C.c_int(), C.c_int()
        pitch, fs = C.c_size_t(), C.c_size_t()
        ptr = C.c_void_p()
        check(lib().orbfe_level_info(self.handle, level, C.byref(w), C.byref(h), C.byref(pitch),
                                     C.byref(ptr), C.byref(fs)), self.handle)
        return w.value, h.value, pitch.value, ptr.value or 0, fs.value

    def build_pyramid(self, d_gray, pitch, frame_stride, n_frames, stream=0):
        check(lib().orbfe_build_pyramid(self.handle, d_gray, pitch, frame_stride, n_frames, stream),
              self.handle)

    def detect_batch(self, n_frames, stream=0):
        check(lib().orbfe_detect_batch(self.handle, n_frames, stream), self.handle)

    def detect_batch_shard(self, n_frames, shard_index, shard_count, stream=0):
        check(lib().orbfe_detect_batch_shard(self.handle, n_frames, shard_index, shard_count, stream),
              self.handle)

    def export_cell_keys(self, n_frames, d_keys, stream=0):
        check(lib().orbfe_export_cell_keys(self.handle, n_frames, d_keys, stream), self.handle)

    def import_cell_keys(self, n_frames, d_keys, stream=0):
        check(lib().orbfe_import_cell_keys(self.handle, n_frames, d_keys, stream), self.handle)

    def describe_batch(self, n_frames, d_records, d_counts, soa=None, stream=0):
        check(lib().orbfe_describe_batch(self.handle, n_frames, d_records, d_counts,
                                         C.byref(soa) if soa is not None else None, stream),
              self.handle)

    def extract(self, d_gray, pitch, frame_stride, n_frames, d_records, d_counts, soa=None,
                stream=0):
        check(lib().orbfe_extract(self.handle, d_gray, pitch, frame_stride, n_frames, d_records,
                                  d_counts, C.byref(soa) if soa is not None else None, stream),
              self.handle)

    def extract_rgb(self, d_rgb, pitch, frame_stride, n_frames, d_records, d_counts, soa=None, stream=0):
        check(lib().orbfe_extract_rgb(self.handle, d_rgb, pitch, frame_stride, n_frames, d_records,
                                      d_counts, C.byref(soa) if soa is not None else None, stream),
              self.handle)

    def match_batch(self, d_records, d_counts, n_frames, mode, window, max_distance, d_idx,
                    d_dist=None, stream=0):
        check(lib().orbfe_match_batch(self.handle, d_records, d_counts, n_frames, mode, window,
                                      max_distance, d_idx, d_dist, stream), self.handle)

    def match_pairs(self, d_records, d_counts, n_frames, first, stride, mode, window, max_distance, d_idx,
                    d_dist=None, stream=0):
        check(lib().orbfe_match_pairs(self.handle, d_records, d_counts, n_frames, first, stride, mode, window,
                                      max_distance, d_idx, d_dist, stream), self.handle)

    def dispatch_info(self, n_frames, mode=1, window=-1):
        """{'pyramid': ..., 'detect': ..., 'describe': ..., 'match': ..., 'match_examines': ...}: the kernels a call runs."""
        buf = C.create_string_buffer(512)
        check(lib().orbfe_dispatch_info(self.handle, n_frames, mode, window, buf, 512), self.handle)
        return dict(kv.split("=", 1) for kv in buf.value.decode().split(";"))

    def selfcheck_steer_table(self):
        """(orientations checked, mismatching sample offsets): the steering table against the arithmetic, exhaustively."""
        n, bad = C.c_ulonglong(0), C.c_ulonglong(0)
        check(lib().orbfe_selfcheck_steer_table(self.handle, C.byref(n), C.byref(bad)), self.handle)
        return n.value, bad.value

    def read_level(self, level, frame=0, stream=0):
        """Copy one pyramid level of one frame to a numpy array [h, w] (harness helper)."""
        w, h, pitch, ptr, fs = self.level_info(level)
        if w == 0 or h == 0:
            return np.zeros((h, w), np.uint8)
        buf = np.empty((h, pitch), np.uint8)
        check(lib().orbfe_memcpy_d2h(buf.ctypes.data, ptr + frame * fs, h * pitch, stream))
        return buf[:, :w].copy()


def _host_array(ptr, dtype, count):
    """numpy view (no copy) of `count` elements of pinned host memory the library owns."""
    dt = np.dtype(dtype)
    buf = (C.c_uint8 * (count * dt.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dt, count=count)


class Ingest:
    """orbfe_ingest (include/orbfe_ingest.h): a ring of pinned host slots around Context.extract (+ match_batch); upload of
    slot s + 1 and download of slot s - 1 overlap the extraction of slot s.  Host arrays are views of the library's pinned
    buffers, not copies."""

    def __init__(self, ctx, frames_per_slot, slots=3, channels=1, match_mode=-1, match_window=-1, match_max_distance=256,
                 download_matches=0):
        self.ctx = ctx  # keeps the context alive
        self.cfg = IngestConfig(slots, frames_per_slot, channels, match_mode, match_window, match_max_distance,
                                download_matches, 0)
        h = C.c_void_p()
        code = lib().orbfe_ingest_create(ctx.handle, C.byref(self.cfg), C.byref(h))
        if code != OK:
            msg = lib().orbfe_ingest_last_error(None)
            raise OrbfeError(code, msg.decode() if msg else "")
        self.handle = h
        self.slots = slots
        self.frames_per_slot = frames_per_slot
        self.frame_bytes = lib().orbfe_ingest_frame_bytes(h)
        self.shape = (ctx.cfg.height, ctx.cfg.width) + ((3,) if channels == 3 else ())

    def _check(self, code):
        if code != OK:
            msg = lib().orbfe_ingest_last_error(self.handle)
            raise OrbfeError(code, msg.decode() if msg else "")

    def close(self):
        if getattr(self, "handle", None):
            lib().orbfe_ingest_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def host_frames(self, slot):
        """[frames_per_slot, H, W(, 3)] u8 view of the slot's pinned input buffer."""
        p = lib().orbfe_ingest_host_frames(self.handle, slot)
        if not p:
            raise OrbfeError(ERR_INVALID_ARG, "no such slot")
        return _host_array(p, np.uint8, self.frames_per_slot * self.frame_bytes).reshape((self.frames_per_slot,) + self.shape)

    def submit(self, slot, n_frames):
        self._check(lib().orbfe_ingest_submit(self.handle, slot, n_frames))

    def submit_from(self, slot, frames):
        """frames: C-contiguous numpy [n, H, W(, 3)] u8 in ordinary (pageable) memory."""
        a = np.ascontiguousarray(frames, dtype=np.uint8)
        n = a.shape[0]
        row = self.shape[1] * (3 if len(self.shape) == 3 else 1)
        self._check(lib().orbfe_ingest_submit_from(self.handle, slot, n, a.ctypes.data, row, self.frame_bytes))

    def ready(self, slot):
        return bool(lib().orbfe_ingest_ready(self.handle, slot))

    def wait(self, slot, n_frames=None):
        """(records [n, cap], counts [n], match_idx or None, match_dist or None) as views of the slot's pinned result buffers."""
        r, c, i, d = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._check(lib().orbfe_ingest_wait(self.handle, slot, C.byref(r), C.byref(c), C.byref(i), C.byref(d)))
        n = self.frames_per_slot if n_frames is None else n_frames
        cap = self.ctx.cap
        rec = _host_array(r.value, KEYPOINT_DTYPE, n * cap).reshape(n, cap)
        cnt = _host_array(c.value, np.int32, n)
        idx = _host_array(i.value, np.int32, max(n - 1, 0) * cap).reshape(max(n - 1, 0), cap) if i.value else None
        dst = _host_array(d.value, np.int32, max(n - 1, 0) * cap).reshape(max(n - 1, 0), cap) if d.value else None
        return rec, cnt, idx, dst

    def device_buffers(self, slot):
        p = [C.c_void_p() for _ in range(5)]
        self._check(lib().orbfe_ingest_device_buffers(self.handle, slot, *[C.byref(x) for x in p]))
        return tuple(x.value or 0 for x in p)

    def compute_stream(self):
        return lib().orbfe_ingest_compute_stream(self.handle) or 0

    def timing(self, slot):
        """{'upload_ms', 'compute_ms', 'download_ms', 'upload_bytes', 'download_bytes'} of the slot's last completed pass."""
        t = [C.c_float() for _ in range(3)]
        b = [C.c_size_t() for _ in range(2)]
        self._check(lib().orbfe_ingest_timing(self.handle, slot, *[C.byref(x) for x in t + b]))
        return dict(upload_ms=t[0].value, compute_ms=t[1].value, download_ms=t[2].value, upload_bytes=b[0].value,
                    download_bytes=b[1].value)
