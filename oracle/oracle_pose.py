"""numpy restatement of the reference's pose-from-matches code (f4).  TEST INFRASTRUCTURE ONLY.

best_fit_transform / nearest_neighbot / icp: src/SlamGpuPipeline/buildStream.cpp:29-188 (Eigen);
process_gyro / process_accel: src/SlamGpuPipeline/SlamGpuPipeline.cpp:179-239.
Parity unpinned at the rounding level: Eigen's JacobiSVD is replaced by numpy.linalg.svd (LAPACK) here
and by a 3x3 Jacobi SVD in the product; tests compare at 1e-9.  The IMU filter's atan2 is the build's
deterministic atan2f (oracle.atan2f), as everywhere else."""
import math

import numpy as np


def best_fit_transform(A, B):
    """:29-85.  A, B: [n, 3].  Returns the 4x4 T (row-major numpy) with B ~= R A + t."""
    A = np.asarray(A, np.float64)
    B = np.asarray(B, np.float64)
    ca, cb = A.sum(0) / len(A), B.sum(0) / len(B)       # :41-47
    H = (A - ca).T @ (B - cb)                           # :54
    U, S, Vt = np.linalg.svd(H)                         # :62-66, singular values descending like Eigen
    R = Vt.T @ U.T                                      # :68
    if np.linalg.det(R) < 0:                            # :70-75
        Vt = Vt.copy()
        Vt[2, :] *= -1
        R = Vt.T @ U.T
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = cb - R @ ca                              # :77
    return T


def nearest_neighbour(src, dst):
    """:96-132: brute force, the distance held in float, running minimum 100, index 0."""
    src = np.asarray(src, np.float64)
    dst = np.asarray(dst, np.float64)
    d = np.sqrt(((src[:, None, :] - dst[None, :, :]) ** 2).sum(-1)).astype(np.float32)
    idx = np.zeros(len(src), np.int64)
    mn = np.full(len(src), np.float32(100))
    for j in range(len(dst)):  # strict <: the first of equal minima wins
        better = d[:, j] < mn
        mn = np.where(better, d[:, j], mn)
        idx = np.where(better, j, idx)
    return mn, idx


def icp(A, B, max_iterations, tolerance):
    """:134-188."""
    A = np.asarray(A, np.float64)
    B = np.asarray(B, np.float64)
    src = A.copy()
    prev_error = 0.0
    for _ in range(max_iterations):
        dist, idx = nearest_neighbour(src, B)            # :158
        T = best_fit_transform(src, B[idx])              # :160-164
        src = src @ T[:3, :3].T + T[:3, 3]               # :166-170
        mean_error = float(dist.astype(np.float64).sum() / len(dist))  # :172 (accumulate(..., 0.0): double)
        if abs(prev_error - mean_error) < tolerance:     # :173 (int tolerance)
            break
        prev_error = mean_error
    return best_fit_transform(A, src)                    # :181


class Imu:
    """SlamGpuPipeline.cpp:179-239 with float32 state."""

    def __init__(self, atan2f):
        self.theta = np.zeros(3, np.float32)
        self.alpha = np.float32(0.98)
        self.first_gyro = self.first_accel = True
        self.last_ts_gyro = 0.0
        self.atan2f = atan2f

    def process_gyro(self, gyro, ts):
        if self.first_gyro:
            self.first_gyro = False
            self.last_ts_gyro = ts
            return
        dt = (ts - self.last_ts_gyro) / 1000.0
        self.last_ts_gyro = ts
        g = [np.float32(float(np.float32(v)) * dt) for v in gyro]  # float *= double
        self.theta[0] = np.float32(self.theta[0] - g[2])
        self.theta[1] = np.float32(self.theta[1] - g[1])
        self.theta[2] = np.float32(self.theta[2] + g[0])

    def process_accel(self, accel):
        a = [np.float32(v) for v in accel]
        az = self.atan2f(a[1], a[2])
        ax = self.atan2f(a[0], np.sqrt(np.float32(np.float32(a[1] * a[1]) + np.float32(a[2] * a[2]))))
        if self.first_accel:
            self.first_accel = False
            self.theta[:] = (ax, np.float32(math.pi), az)
        else:
            one_m = np.float32(np.float32(1) - self.alpha)
            self.theta[0] = np.float32(np.float32(self.theta[0] * self.alpha) + np.float32(ax * one_m))
            self.theta[2] = np.float32(np.float32(self.theta[2] * self.alpha) + np.float32(az * one_m))
