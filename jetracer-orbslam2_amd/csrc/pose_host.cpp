// pose_host.cpp -- include/orbfe_pose.h: rigid fit of matched 3-D points, the reference's ICP loop and
// its IMU complementary filter (src/SlamGpuPipeline/buildStream.cpp:29-188, SlamGpuPipeline.cpp:179-239).
// Host code only; part of liborbfe.so.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/orbfe.h"
#include "../../include/orbfe_math.h"
#include "../../include/orbfe_pose.h"

namespace {

// 3 x 3 SVD by one-sided Jacobi: H = U diag(s) V^T, singular values sorted descending (as Eigen's
// JacobiSVD returns them; the det < 0 repair of best_fit_transform relies on the LAST one being the smallest).
void svd3(const double H[3][3], double U[3][3], double s[3], double V[3][3])
{
    double W[3][3]; // columns are rotated until mutually orthogonal: W = H V
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            W[i][j] = H[i][j];
            V[i][j] = i == j ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double a = 0, b = 0, c = 0;
                for (int i = 0; i < 3; i++) {
                    a += W[i][p] * W[i][p];
                    b += W[i][q] * W[i][q];
                    c += W[i][p] * W[i][q];
                }
                off += c * c;
                if (std::fabs(c) <= 1e-300 || std::fabs(c) <= 1e-17 * std::sqrt(a * b)) continue;
                const double zeta = (b - a) / (2.0 * c);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
                for (int i = 0; i < 3; i++) {
                    const double wp = W[i][p], wq = W[i][q];
                    W[i][p] = cs * wp - sn * wq;
                    W[i][q] = sn * wp + cs * wq;
                    const double vp = V[i][p], vq = V[i][q];
                    V[i][p] = cs * vp - sn * vq;
                    V[i][q] = sn * vp + cs * vq;
                }
            }
        if (off == 0.0) break;
    }
    int order[3] = {0, 1, 2};
    double n2[3];
    for (int j = 0; j < 3; j++) n2[j] = W[0][j] * W[0][j] + W[1][j] * W[1][j] + W[2][j] * W[2][j];
    for (int i = 0; i < 2; i++)
        for (int j = i + 1; j < 3; j++)
            if (n2[order[j]] > n2[order[i]]) {
                const int t = order[i];
                order[i] = order[j];
                order[j] = t;
            }
    double Vs[3][3];
    for (int j = 0; j < 3; j++) {
        const int o = order[j];
        s[j] = std::sqrt(n2[o]);
        for (int i = 0; i < 3; i++) {
            Vs[i][j] = V[i][o];
            U[i][j] = s[j] > 0 ? W[i][o] / s[j] : 0.0;
        }
    }
    memcpy(V, Vs, sizeof(Vs));
    // rank-deficient input: complete U to an orthonormal basis (cross products), so that V U^T is orthogonal
    if (!(s[1] > 1e-300 * (s[0] + 1.0))) {
        // pick any unit vector orthogonal to column 0
        const double *u0 = &U[0][0];
        double a[3] = {U[0][0], U[1][0], U[2][0]};
        (void)u0;
        if (!(s[0] > 0)) { a[0] = 1; a[1] = 0; a[2] = 0; U[0][0] = 1; U[1][0] = 0; U[2][0] = 0; }
        int k = std::fabs(a[0]) < std::fabs(a[1]) ? (std::fabs(a[0]) < std::fabs(a[2]) ? 0 : 2) : (std::fabs(a[1]) < std::fabs(a[2]) ? 1 : 2);
        double e[3] = {0, 0, 0};
        e[k] = 1;
        double b[3] = {a[1] * e[2] - a[2] * e[1], a[2] * e[0] - a[0] * e[2], a[0] * e[1] - a[1] * e[0]};
        const double nb = std::sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
        for (int i = 0; i < 3; i++) U[i][1] = b[i] / nb;
    }
    if (!(s[2] > 1e-300 * (s[0] + 1.0))) {
        U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
        U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
        U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
    }
}

double det3(const double R[3][3])
{
    return R[0][0] * (R[1][1] * R[2][2] - R[1][2] * R[2][1]) - R[0][1] * (R[1][0] * R[2][2] - R[1][2] * R[2][0]) +
           R[0][2] * (R[1][0] * R[2][1] - R[1][1] * R[2][0]);
}

void fit(const double *A, const double *B, int n, double T[16])
{
    double ca[3] = {0, 0, 0}, cb[3] = {0, 0, 0};
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            ca[k] += A[3 * i + k];
            cb[k] += B[3 * i + k];
        }
    for (int k = 0; k < 3; k++) {
        ca[k] /= n;
        cb[k] /= n;
    }
    double H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}; // AA^T * BB
    for (int i = 0; i < n; i++)
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) H[r][c] += (A[3 * i + r] - ca[r]) * (B[3 * i + c] - cb[c]);
    double U[3][3], s[3], V[3][3], R[3][3];
    svd3(H, U, s, V);
    auto make_R = [&]() { // R = V * U^T
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) R[r][c] = V[r][0] * U[c][0] + V[r][1] * U[c][1] + V[r][2] * U[c][2];
    };
    make_R();
    if (det3(R) < 0) { // Vt.block<1,3>(2,0) *= -1, buildStream.cpp:73-78
        for (int r = 0; r < 3; r++) V[r][2] = -V[r][2];
        make_R();
    }
    for (int i = 0; i < 16; i++) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T[4 * c + r] = R[r][c];
        T[12 + r] = cb[r] - (R[r][0] * ca[0] + R[r][1] * ca[1] + R[r][2] * ca[2]);
    }
}

} // namespace

extern "C" {

int orbfe_best_fit_transform(const double *A, const double *B, int n, double T[16])
{
    if (!A || !B || !T || n < 3) return ORBFE_ERR_INVALID_ARG;
    fit(A, B, n, T);
    return ORBFE_OK;
}

int orbfe_icp(const double *A, const double *B, int n, int max_iterations, int tolerance, double T[16])
{
    if (!A || !B || !T || n < 3 || max_iterations < 0) return ORBFE_ERR_INVALID_ARG;
    std::vector<double> src(A, A + 3 * (size_t)n), chosen(3 * (size_t)n);
    std::vector<float> dist(n);
    double prev_error = 0, step[16];
    for (int it = 0; it < max_iterations; it++) {
        // nearest_neighbot, :96-132: float distances, running minimum starts at 100, index at 0
        for (int i = 0; i < n; i++) {
            float mn = 100;
            int idx = 0;
            for (int j = 0; j < n; j++) {
                const double dx = src[3 * i] - B[3 * j], dy = src[3 * i + 1] - B[3 * j + 1], dz = src[3 * i + 2] - B[3 * j + 2];
                const float d = (float)std::sqrt(dx * dx + dy * dy + dz * dz);
                if (d < mn) {
                    mn = d;
                    idx = j;
                }
            }
            dist[i] = mn;
            for (int k = 0; k < 3; k++) chosen[3 * i + k] = B[3 * idx + k];
        }
        fit(src.data(), chosen.data(), n, step);
        for (int i = 0; i < n; i++) { // src = T * src
            const double x = src[3 * i], y = src[3 * i + 1], z = src[3 * i + 2];
            for (int r = 0; r < 3; r++) src[3 * i + r] = step[r] * x + step[4 + r] * y + step[8 + r] * z + step[12 + r];
        }
        double mean_error = 0;
        for (int i = 0; i < n; i++) mean_error += dist[i];
        mean_error /= n;
        if (std::fabs(prev_error - mean_error) < tolerance) break;
        prev_error = mean_error;
    }
    fit(A, src.data(), n, T);
    return ORBFE_OK;
}

void orbfe_imu_init(orbfe_imu *s)
{
    if (!s) return;
    memset(s, 0, sizeof(*s));
    s->alpha = 0.98f; // SlamGpuPipeline.h:73
    s->first_gyro = s->first_accel = 1;
}

void orbfe_imu_process_gyro(orbfe_imu *s, const float gyro[3], double ts)
{
    if (!s || !gyro) return;
    if (s->first_gyro) {
        s->first_gyro = 0;
        s->last_ts_gyro = ts;
        return;
    }
    const double dt = (ts - s->last_ts_gyro) / 1000.0;
    s->last_ts_gyro = ts;
    // float *= double: the product is formed in double and narrowed (:199-201)
    const float gx = (float)((double)gyro[0] * dt), gy = (float)((double)gyro[1] * dt), gz = (float)((double)gyro[2] * dt);
    s->theta[0] -= gz;
    s->theta[1] -= gy;
    s->theta[2] += gx;
}

void orbfe_imu_process_accel(orbfe_imu *s, const float accel[3])
{
    if (!s || !accel) return;
    // float arithmetic (:217-218); atan2f is the build's deterministic one (the reference's is libm / libdevice:
    // unpinned at the ulp level, as for the orientation angle)
    const float az = orbfe_atan2f(accel[1], accel[2]);
    const float ax = orbfe_atan2f(accel[0], sqrtf(accel[1] * accel[1] + accel[2] * accel[2]));
    if (s->first_accel) {
        s->first_accel = 0;
        s->theta[0] = ax;
        s->theta[1] = (float)3.14159265358979323846; // CUDART_PI_D narrowed to float (:227)
        s->theta[2] = az;
    } else {
        s->theta[0] = s->theta[0] * s->alpha + ax * (1 - s->alpha);
        s->theta[2] = s->theta[2] * s->alpha + az * (1 - s->alpha);
    }
}

} // extern "C"
