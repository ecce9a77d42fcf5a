/* orbfe_math.h -- deterministic single-precision helpers shared by the HIP kernels and
 * by the CPU oracle, so that both sides produce the same bits.
 *
 * Why this exists: the reference calls CUDA libdevice atan2f / cosf / sinf
 * (src/cuda/orb.cu:140, :45, :46).  Those are few-ulp approximations whose bits cannot be
 * reproduced off NVIDIA hardware, and glibc / ROCm OCML differ from them and from each
 * other.  The build therefore owns ONE definition of each routine, written only with
 * IEEE-754 single operations (+ - * /, compare, convert) that are correctly rounded on
 * x86-64 and on gfx950 alike.  Every translation unit that includes this header MUST be
 * compiled with -ffp-contract=off (no fused multiply-add may be formed); the Makefiles do
 * so and tests/test_kat.py checks known bit patterns on the CPU, tests/test_gpu_parity.py
 * on the GPU.  Accuracy versus the true function is <= ~3 ulp (tests compare with libm).
 *
 * The algorithms are the classic single-precision Cephes forms (argument reduction to
 * [0, tan(pi/8)] plus a degree-4 polynomial in x^2 for atan; pi/4 octant reduction with a
 * 3-term Cody-Waite split plus degree-3 polynomials for sin/cos).
 *
 * Plain C99 / C++ / HIP.  No dependencies.
 */
#ifndef ORBFE_MATH_H
#define ORBFE_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define ORBFE_HD __host__ __device__ static inline
#else
#define ORBFE_HD static inline
#endif

#if defined(__clang__)
#define ORBFE_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define ORBFE_NO_CONTRACT
#endif

#define ORBFE_PI_F 3.14159274101257324f     /* (float)pi            */
#define ORBFE_PIO2_F 1.57079637050628662f   /* (float)(pi/2)        */
#define ORBFE_PIO4_F 0.785398185253143311f  /* (float)(pi/4)        */
/* (float)(CUDART_PI_F / 180.f) with CUDART_PI_F = 3.141592654f (src/cuda_common.h:49-51,
 * src/cuda/orb.cu:42): a float/float division rounded to float. */
#define ORBFE_DEG2RAD_F (3.141592654f / 180.f)

ORBFE_HD float orbfe_fabsf(float v) { return v < 0.0f ? -v : v; }

/* atan(x) for x >= 0. */
ORBFE_HD float orbfe_atanf_nonneg(float x)
{
    ORBFE_NO_CONTRACT
    float y0, z, p;
    if (x > 2.414213562373095f) { /* tan(3pi/8) */
        y0 = ORBFE_PIO2_F;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) { /* tan(pi/8) */
        y0 = ORBFE_PIO4_F;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y0 = 0.0f;
    }
    z = x * x;
    p = 8.05374449538e-2f * z;
    p = p - 1.38776856032e-1f;
    p = p * z;
    p = p + 1.99777106478e-1f;
    p = p * z;
    p = p - 3.33329491539e-1f;
    p = p * z;
    p = p * x;
    p = p + x;
    return y0 + p;
}

/* atan2f(y, x) in (-pi, pi]; atan2f(0,0) = 0. */
ORBFE_HD float orbfe_atan2f(float y, float x)
{
    ORBFE_NO_CONTRACT
    float q, a, w;
    if (x == 0.0f) {
        if (y > 0.0f) return ORBFE_PIO2_F;
        if (y < 0.0f) return -ORBFE_PIO2_F;
        return 0.0f;
    }
    if (y == 0.0f) return x > 0.0f ? 0.0f : ORBFE_PI_F;
    q = y / x;
    a = orbfe_atanf_nonneg(orbfe_fabsf(q));
    if (q < 0.0f) a = -a;
    w = 0.0f;
    if (x < 0.0f) w = (y < 0.0f) ? -ORBFE_PI_F : ORBFE_PI_F;
    return w + a;
}

/* sin and cos of x, |x| < 8192. */
ORBFE_HD void orbfe_sincosf(float xx, float *s_out, float *c_out)
{
    ORBFE_NO_CONTRACT
    float x = orbfe_fabsf(xx);
    unsigned j = (unsigned)(x * 1.27323954473516f); /* floor(x / (pi/4)) */
    float y = (float)j;
    float z, ps, pc, s, c;
    if (j & 1u) {
        j += 1u;
        y = y + 1.0f;
    }
    j &= 7u;
    /* x - y*pi/4 in three exact pieces */
    x = x - y * 0.78515625f;
    x = x - y * 2.4187564849853515625e-4f;
    x = x - y * 3.77489497744594108e-8f;
    z = x * x;

    ps = -1.9515295891e-4f * z;
    ps = ps + 8.3321608736e-3f;
    ps = ps * z;
    ps = ps - 1.6666654611e-1f;
    ps = ps * z;
    ps = ps * x;
    ps = ps + x;

    pc = 2.443315711809948e-5f * z;
    pc = pc - 1.388731625493765e-3f;
    pc = pc * z;
    pc = pc + 4.166664568298827e-2f;
    pc = pc * z;
    pc = pc * z;
    pc = pc - 0.5f * z;
    pc = pc + 1.0f;

    /* octant j in {0,2,4,6}: angle = r + j*pi/4 */
    switch (j) {
    case 0: s = ps; c = pc; break;
    case 2: s = pc; c = -ps; break;
    case 4: s = -ps; c = -pc; break;
    default: s = -pc; c = ps; break; /* 6 */
    }
    if (xx < 0.0f) s = -s;
    *s_out = s;
    *c_out = c;
}

/* atan(x), any sign; tan(x) = sin / cos of orbfe_sincosf (one IEEE division), |x| < 8192. */
ORBFE_HD float orbfe_atanf(float x) { return x < 0.0f ? -orbfe_atanf_nonneg(-x) : orbfe_atanf_nonneg(x); }
ORBFE_HD float orbfe_tanf(float x)
{
    float s, c;
    orbfe_sincosf(x, &s, &c);
    return s / c;
}

/* The f-theta branch of librealsense's project_point_to_pixel as the reference compiles it
 * (src/cuda/cuda-align.cu:44-50, src/cuda/post_processing.cu:32-38):
 *     float r = sqrtf(x*x + y*y);
 *     float rd = (float)(1.0f / coeffs[0] * atan(2 * r * tan(coeffs[0] / 2.0f)));
 *     x *= rd / r;  y *= rd / r;
 * Every operand is a float and the file is C++ (nvcc), so `tan` and `atan` resolve to the float overloads, i.e.
 * libdevice's tanf / atanf: the whole expression is single precision.  Decision: the same operations in the same order
 * with the build's own deterministic atanf / tanf above and a correctly rounded sqrtf; PARITY UNPINNED at the ulp
 * level (libdevice's bits are not reproducible here), oracle == HIP bit for bit.  r == 0 gives 0 / 0 = NaN exactly as
 * in the reference (the pixel then converts to 0 with cvt.rzi's NaN rule). */
ORBFE_HD void orbfe_ftheta_distort(float *x, float *y, float c0)
{
    ORBFE_NO_CONTRACT
    const float xx = *x, yy = *y;
    const float r = __builtin_sqrtf(xx * xx + yy * yy);
    const float t = orbfe_tanf(c0 / 2.0f);
    const float rd = 1.0f / c0 * orbfe_atanf(2 * r * t);
    *x = xx * (rd / r);
    *y = yy * (rd / r);
}

/* Round to nearest, ties to even, then to int: the meaning of CUDA __float2int_rn
 * (src/cuda/orb.cu:13-14).  rintf under the default rounding mode; v_rndne_f32 on gfx950. */
ORBFE_HD int orbfe_rn_int(float v) { return (int)__builtin_rintf(v); }

/* Does the 16-bit ring mask m hold a cyclic run of >= arc ones (9 <= arc <= 15)?
 * Closed form of the reference's LUT predicate (src/cuda/fast.cu:11-32); test_kat.py checks
 * it against the oracle's literal restatement for all 65536 masks and arc 9..12. */
ORBFE_HD int orbfe_has_arc(uint32_t m, int arc)
{
    uint32_t t = (m & 0xFFFFu) | (m << 16);
    int have = 8;
    t &= t >> 1; /* runs >= 2 */
    t &= t >> 2; /* runs >= 4 */
    t &= t >> 4; /* runs >= 8: bit i set iff bits i..i+7 all set */
    /* extend from 8 to arc (<= 16) */
    if (arc - have >= 4) { t &= t >> 4; have += 4; }
    if (arc - have >= 2) { t &= t >> 2; have += 2; }
    if (arc - have >= 1) { t &= t >> 1; have += 1; }
    /* a run may start at any of the 16 ring positions: bits 0..15 of the doubled mask */
    return (t & 0xFFFFu) != 0u;
}

ORBFE_HD uint32_t orbfe_bitrev5(uint32_t v)
{
    return ((v & 1u) << 4) | ((v & 2u) << 2) | (v & 4u) | ((v & 8u) >> 2) | ((v & 16u) >> 4);
}

#endif /* ORBFE_MATH_H */
