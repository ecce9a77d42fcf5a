"""f4 (include/orbfe_pose.h), host code: rigid fit, the reference's ICP loop and its IMU filter against the
numpy restatement in oracle/oracle_pose.py.  Tolerance 1e-9 on the transforms (the reference's Eigen
JacobiSVD, LAPACK here and the product's own 3x3 Jacobi SVD agree to rounding, not bit for bit); the IMU
filter is float arithmetic with the build's deterministic atan2f and is compared bit for bit."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-9


@pytest.fixture(scope="module")
def L():
    import orbfe
    if not os.path.exists(orbfe.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return orbfe.lib()


def _rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _fit(L, A, B):
    T = (C.c_double * 16)()
    A = np.ascontiguousarray(A, np.float64)
    B = np.ascontiguousarray(B, np.float64)
    assert L.orbfe_best_fit_transform(A.ctypes.data, B.ctypes.data, len(A), T) == 0
    return np.array(T).reshape(4, 4).T  # column-major -> numpy


def test_header_and_binding_agree(L):
    import orbfe
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "orbfe_pose.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(orbfe_[a-z0-9_]+)\s*\(", text)))
    assert declared == sorted(orbfe.POSE_EXPORTS)
    for name in declared:
        assert hasattr(L, name)
    assert C.sizeof(orbfe.Imu) == 32


@pytest.mark.parametrize("n,noise", [(3, 0.0), (10, 0.0), (300, 0.01), (2000, 0.5)])
def test_best_fit_transform(L, n, noise):
    import oracle_pose
    rng = np.random.default_rng(n)
    A = rng.normal(size=(n, 3)) * [400, 300, 2000]
    R, t = _rot(rng), rng.normal(size=3) * 50
    B = A @ R.T + t + rng.normal(size=(n, 3)) * noise
    T = _fit(L, A, B)
    ref = oracle_pose.best_fit_transform(A, B)
    np.testing.assert_allclose(T, ref, rtol=0, atol=TOL * max(1.0, np.abs(ref).max()))
    assert abs(np.linalg.det(T[:3, :3]) - 1) < 1e-12 and np.allclose(T[3], [0, 0, 0, 1])
    if noise == 0.0:
        np.testing.assert_allclose(T[:3, :3], R, atol=1e-9)
        np.testing.assert_allclose(T[:3, 3], t, atol=1e-6)


def test_best_fit_transform_reflection_and_degenerate_inputs(L):
    """det(V U^T) < 0 (mirrored target): the third row of V^T is negated (buildStream.cpp:70-75); coplanar and
    collinear inputs still give a proper rotation."""
    import oracle_pose
    rng = np.random.default_rng(5)
    A = rng.normal(size=(50, 3))
    B = A * [1, 1, -1]  # a reflection: the best ROTATION is what both sides must return
    T, ref = _fit(L, A, B), oracle_pose.best_fit_transform(A, B)
    np.testing.assert_allclose(T, ref, atol=1e-9)
    assert np.linalg.det(T[:3, :3]) > 0.999999
    planar = np.concatenate([rng.normal(size=(40, 2)), np.zeros((40, 1))], 1)
    R = _rot(rng)
    T = _fit(L, planar, planar @ R.T)
    np.testing.assert_allclose(T[:3, :3], R, atol=1e-9)
    line = np.outer(rng.normal(size=20), [1.0, 2.0, -0.5])
    T = _fit(L, line, line + [1, 2, 3])
    Rl = T[:3, :3]
    np.testing.assert_allclose(Rl @ Rl.T, np.eye(3), atol=1e-9)
    assert abs(np.linalg.det(Rl) - 1) < 1e-9
    np.testing.assert_allclose(line @ Rl.T + T[:3, 3], line + [1, 2, 3], atol=1e-9)
    assert L.orbfe_best_fit_transform(A.ctypes.data, B.ctypes.data, 2, (C.c_double * 16)()) != 0


@pytest.mark.parametrize("tolerance,iters", [(0, 12), (1, 12), (0, 0)])
def test_icp_follows_the_reference_loop(L, tolerance, iters):
    import oracle_pose
    rng = np.random.default_rng(8)
    n = 120
    B = rng.uniform(-40, 40, size=(n, 3))
    a = 0.05
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    A = (B - [0.3, -0.2, 0.1]) @ R  # B = R A + t, unknown correspondence order
    A = A[rng.permutation(n)]
    A[:3] += 500  # farther than 100 from every target: pairs with target 0 (the running minimum starts at 100)
    T = (C.c_double * 16)()
    assert L.orbfe_icp(A.ctypes.data, B.ctypes.data, n, iters, tolerance, T) == 0
    got = np.array(T).reshape(4, 4).T
    ref = oracle_pose.icp(A, B, iters, tolerance)
    np.testing.assert_allclose(got, ref, atol=1e-8)
    if iters == 0:
        np.testing.assert_allclose(got, np.eye(4), atol=1e-12)


def test_imu_complementary_filter_bits(L, oracle_mod):
    import orbfe
    import oracle_pose
    rng = np.random.default_rng(3)
    s = orbfe.Imu()
    L.orbfe_imu_init(C.byref(s))
    ref = oracle_pose.Imu(oracle_mod.atan2f)
    assert abs(s.alpha - 0.98) < 1e-7 and s.first_gyro == 1 and s.first_accel == 1
    ts = 1000.0
    for k in range(200):
        if k % 3 == 0:
            acc = (rng.normal(size=3) * [0.5, 0.5, 0.5] + [0.1, -9.7, 0.8]).astype(np.float32)
            L.orbfe_imu_process_accel(C.byref(s), (C.c_float * 3)(*acc))
            ref.process_accel(acc)
        else:
            gy = (rng.normal(size=3) * 0.2).astype(np.float32)
            ts += float(rng.uniform(2, 6))
            L.orbfe_imu_process_gyro(C.byref(s), (C.c_float * 3)(*gy), ts)
            ref.process_gyro(gy, ts)
        got = np.array(list(s.theta), np.float32)
        assert got.view(np.uint32).tolist() == ref.theta.view(np.uint32).tolist(), "step %d" % k
    assert abs(float(s.theta[1]) - np.pi) < 0.5  # started at pi (SlamGpuPipeline.cpp:227), drifted by the gyro only


@pytest.mark.parametrize("n,noise", [(4, 0.0), (50, 0.02), (1000, 0.3)])
def test_best_fit_transform_against_scipy_kabsch(L, n, noise):
    """A third, independent implementation: SciPy's Kabsch solver (Rotation.align_vectors) on the centred point sets.
    The reference's best_fit_transform (buildStream.cpp:29-90) is that algorithm (SVD of the cross-covariance with the
    reflection fix); the product's own 3x3 Jacobi SVD must land on the same rotation and translation."""
    Rotation = pytest.importorskip("scipy.spatial.transform").Rotation
    rng = np.random.default_rng(100 + n)
    A = rng.normal(size=(n, 3)) * [300, 200, 1500]
    R, t = _rot(rng), rng.normal(size=3) * 20
    B = A @ R.T + t + rng.normal(size=(n, 3)) * noise
    T = _fit(L, A, B)
    ca, cb = A.mean(0), B.mean(0)
    rot, _ = Rotation.align_vectors(B - cb, A - ca)  # the rotation that takes A - ca onto B - cb
    Rs = rot.as_matrix()
    assert np.abs(T[:3, :3] - Rs).max() < 1e-8
    assert np.abs(T[:3, 3] - (cb - Rs @ ca)).max() < 1e-6
