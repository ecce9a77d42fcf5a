/* orbfe_pose.h -- f4 of SURVEY.md 8f: what consumes the matcher's output in the reference, as host code
 * (part of liborbfe.so, no HIP): the rigid fit of matched 3-D point lists and the ICP loop around it
 * (src/SlamGpuPipeline/buildStream.cpp:29-188; the reference's only call is commented out at :572), and
 * the IMU complementary filter that produces slam_frame_t::theta (src/SlamGpuPipeline/SlamGpuPipeline.cpp
 * :179-239).  The GPU part of this row, kernel_reproject_prev_points, is orbfe_reproject_points in orbfe.h.
 *
 * Matrices are 16 doubles, COLUMN-major, as Eigen::Matrix4d::data() gives them.  Point lists are n x 3
 * doubles, row-major: exactly the double3 arrays orbfe_match_compact writes.
 *
 * Parity: the reference computes the fit with Eigen's JacobiSVD, whose bits are not reproducible without
 * Eigen; this build uses its own one-sided Jacobi SVD of the 3 x 3 covariance.  The result is the same
 * rotation and translation up to rounding; tests compare with a numpy SVD restatement at 1e-9.
 */
#ifndef ORBFE_POSE_H
#define ORBFE_POSE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* best_fit_transform, buildStream.cpp:29-85: T (4 x 4) with B ~= R A + t in the least-squares sense;
 * centroids, H = AA^T BB, SVD, R = V U^T, third row of V^T negated when det R < 0.  n >= 3. */
int orbfe_best_fit_transform(const double *A, const double *B, int n, double T[16]);

/* icp, buildStream.cpp:134-188, including its quirks: nearest neighbours by brute force with the distance
 * held in FLOAT and the running minimum starting at 100 (a point farther than 100 from every target pairs
 * with target 0, :103-121); `tolerance` is an int compared with |prev_error - mean_error| (:172); the
 * returned transform is best_fit_transform(A, moved A) (:181). */
int orbfe_icp(const double *A, const double *B, int n, int max_iterations, int tolerance, double T[16]);

/* process_gyro / process_accel, SlamGpuPipeline.cpp:179-239; state = the members they touch
 * (SlamGpuPipeline.h: theta, firstGyro, firstAccel, last_ts_gyro, alpha = 0.98). */
typedef struct orbfe_imu {
    float theta[3]; /* x, y, z */
    float alpha;
    double last_ts_gyro;
    int32_t first_gyro, first_accel;
} orbfe_imu;
void orbfe_imu_init(orbfe_imu *s);
void orbfe_imu_process_gyro(orbfe_imu *s, const float gyro[3], double ts_ms);
void orbfe_imu_process_accel(orbfe_imu *s, const float accel[3]);

#ifdef __cplusplus
}
#endif
#endif /* ORBFE_POSE_H */
