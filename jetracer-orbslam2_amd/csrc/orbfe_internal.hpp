// orbfe_internal.hpp -- shared by the HIP translation units of liborbfe.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/orbfe.h"
#include "../../include/orbfe_math.h"
#include "../../include/orbfe_pattern.h"

namespace orbfe {

constexpr int kWave = 64; // gfx950 wavefront

// ---- error plumbing: status codes + per-thread / per-context message ------------------
void set_thread_error(const char *fmt, ...);
const char *thread_error();

#define ORBFE_HIP_TRY(ctx_err, expr)                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            orbfe::format_error(ctx_err, "%s failed: %s (%s:%d)", #expr,                    \
                                hipGetErrorString(e_), __FILE__, __LINE__);                 \
            return ORBFE_ERR_HIP;                                                           \
        }                                                                                   \
    } while (0)

void format_error(char *ctx_err, const char *fmt, ...);

// ---- NMS tie-break in closed form ---------------------------------------------------
// The reference picks, per cell, the maximum response with ties decided by the structure
// of its kernel (src/cuda/nms.cu:105-108, :188, :201-212, :230-244, :246): lower pyramid
// level first, then lower warp id, then lower bit-reversed lane id, then the earlier row
// iteration of the owning thread.  We pack all of it into one
// 27-bit key whose unsigned maximum IS that winner, so the per-cell reduction becomes an
// order-independent max (wave reduction or atomicMax) with a deterministic result:
//     key = score << 15 | (7 - level) << 12 | (4095 - rank)
//     rank = ((warp_id * 32 + bitrev5(lane_id)) << 5) | row_iteration
// score <= 16 * 255 = 4080 < 2^12; level <= 6; rank < 2^12 for every cell size <= 64.
struct NmsGeom {
    int c;    // cell size at this level (power of two)
    int by;   // thread rows of the reference block: max(1, min(128 / c, c))
};

__host__ __device__ inline NmsGeom nms_geom(int cell0, int level)
{
    NmsGeom g;
    g.c = cell0 >> level;
    // 128 / c as a shift (c is a power of two, 1 .. 64: a runtime division costs ~40 instructions per workgroup)
    int t = 128 >> (g.c > 0 ? 31 - __builtin_clz((unsigned)g.c) : 0);
    int m = t < g.c ? t : g.c;
    g.by = m > 1 ? m : 1;
    return g;
}

// floor(log2(v)) for v >= 1 (cell sizes and block rows are powers of two)
__host__ __device__ inline int ilog2(int v)
{
    return 31 - __builtin_clz((unsigned)v);
}

__host__ __device__ inline uint32_t nms_key(int score, int level, int x, int y, int cell0)
{
    const NmsGeom g = nms_geom(cell0, level);
    // c and by are powers of two: divisions become shifts / masks
    const int lc = ilog2(g.c), lby = ilog2(g.by);
    const int cy = y >> lc;
    const int tx = x & (g.c - 1);
    int yoff = 3 - (cy << lc);
    yoff = yoff > 0 ? yoff : 0;
    const int r = y - (cy << lc) - yoff;
    const int ty = r & (g.by - 1), k = r >> lby;
    const int tid = tx + (ty << lc);
    const uint32_t rank = (((uint32_t)(tid >> 5) * 32u + orbfe_bitrev5((uint32_t)tid & 31u)) << 5) |
                          (uint32_t)k;
    return ((uint32_t)score << 15) | ((uint32_t)(7 - level) << 12) | (4095u - rank);
}

__host__ __device__ inline void nms_decode(uint32_t key, int cell_x, int cell_y, int cell0,
                                           int *score, int *level, int *x, int *y)
{
    *score = (int)(key >> 15);
    if (*score == 0) { // empty cell: SURVEY.md Appendix A, Q5
        *level = 0;
        *x = 0;
        *y = 0;
        return;
    }
    int l = 7 - (int)((key >> 12) & 7u);
    uint32_t rank = 4095u - (key & 4095u);
    int k = (int)(rank & 31u);
    uint32_t wl = rank >> 5;
    int tid = (int)((wl >> 5) * 32u + orbfe_bitrev5(wl & 31u));
    NmsGeom g = nms_geom(cell0, l);
    int tx = tid & (g.c - 1), ty = tid >> ilog2(g.c); // c is a power of two
    int yoff = 3 - g.c * cell_y;
    yoff = yoff > 0 ? yoff : 0;
    *level = l;
    *x = (g.c * cell_x + tx) << l;
    *y = (g.c * cell_y + yoff + ty + k * g.by) << l;
}

// ---- FAST ring (src/cuda/fast.cu:41-96), index i -> (dx, dy), +y down ------------------
__host__ __device__ constexpr int ring_dx(int i)
{
    return i == 0 ? 0 : i == 1 ? -1 : i == 2 ? -2 : i == 3 ? -3 : i == 4 ? -3 : i == 5 ? -3
         : i == 6 ? -2 : i == 7 ? -1 : i == 8 ? 0 : i == 9 ? 1 : i == 10 ? 2 : i == 11 ? 3
         : i == 12 ? 3 : i == 13 ? 3 : i == 14 ? 2 : 1;
}
__host__ __device__ constexpr int ring_dy(int i)
{
    return i == 0 ? 3 : i == 1 ? 3 : i == 2 ? 2 : i == 3 ? 1 : i == 4 ? 0 : i == 5 ? -1
         : i == 6 ? -2 : i == 7 ? -3 : i == 8 ? -3 : i == 9 ? -3 : i == 10 ? -2 : i == 11 ? -1
         : i == 12 ? 0 : i == 13 ? 1 : i == 14 ? 2 : 3;
}

// ---- context -----------------------------------------------------------------------
constexpr int kMaxLevels = 16;
constexpr int kTileW = 64, kTileH = 64; // detection tile (pixels of one level)

struct LevelInfo {
    int w, h, pitch;
    size_t offset; // bytes from the frame base inside the pyramid buffer
};

struct TileDesc { // one detection workgroup
    int16_t level, tx, ty, pad;
};

struct DeviceGeom { // passed by value to kernels
    int W, H;
    int L;        // levels built
    int Ld;       // levels detected on
    int cell;     // level-0 cell
    int cells_x, cells_y, K;
    int cap;      // records per frame
    int threshold, arc;
    int max_features;
    int angle_in_radians;
    int descriptor_level; // EXT iv: describe on the keypoint's own pyramid level
    // per launch (with_frames): frames in this launch, and whether the grid is the 8-frames-per-row form of
    // frame_item() (device_common.hpp)
    int n_frames, grid8;
    size_t frame_stride;
    LevelInfo lv[kMaxLevels];
};

// steer_table.cpp: orientation -> rotated rBRIEF sample offsets for LDS tiles of row pitch `pitch`; *offsets =
// [intervals = breaks + 1][lane 64][8] int16 (P, Q of rounds 0..3), interval of an orientation = number of break
// points <= it, *central = the interval of orientation 0
void build_steer_table(int pitch, std::vector<float> *breaks, std::vector<int16_t> *offsets, int *central, uint64_t sched_mask[4]);
// what the tile describe kernel needs of it (null table: compute the offsets instead)
struct SteerArgs {
    const float *breaks;
    const uint4 *table;
    int n_breaks, central;
    uint64_t sched_mask[4]; // lanes whose schedule bit 0..3 is set (steer_table.cpp)
};

// match_mfma.hip: 256-bit brute-force matching on the matrix cores; pair k = frames
// (first + k * stride, first + k * stride + 1), results at k * cap
constexpr int kMmaMaxKeypoints = 16384;
void launch_match_mfma(const orbfe_keypoint *d_records, const int32_t *d_counts, int n_frames, int n_pairs, int first,
                       int stride, int cap, int capP, int max_dist, uint4 *mexp, float *mkey, int form, int32_t *d_idx,
                       int32_t *d_dist, hipStream_t stream);
// form: 0 = by call size, 1 = match_expand_kernel + match_mfma_kernel, 2 = match_tile_kernel (orbfe_ctx::match_form)
bool match_mfma_uses_tile(int n_pairs, int capP, int form);

// batch_kernels.hip: orbfe_detect of the stage API on the fused tile kernel (one launch for all levels,
// scores also written to the caller's response maps); integer threshold, dword-aligned levels; corners are
// decided by the caller's table d_lut, whatever it holds
int launch_detect_stage(const orbfe_pyramid_level *lv, int n_levels, int threshold, const uint8_t *d_lut, float *d_pos,
                        float *d_score, int *d_level, hipStream_t stream);

// a per-frame count read from a caller's device buffer, made safe to index with: a stale or corrupt
// count must not walk past the frame's cap records
__host__ __device__ inline int clamp_count(int n, int cap)
{
    return n < 0 ? 0 : (n > cap ? cap : n);
}

// Makes `device` current for the duration of a call and restores the caller's device on return.
struct DeviceScope {
    int prev = -1;
    bool ok = true;
    explicit DeviceScope(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) ok = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceScope()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

// A small struct the caller hands over by pointer (rs2_intrinsics / rs2_extrinsics shaped): host memory is read in place;
// DEVICE memory -- what the reference passes, _d_depth_intrinsics / _d_rgb_intrinsics / _d_depth_rgb_extrinsics of
// SlamGpuPipeline.cpp:53-55, uploaded once -- is copied to `tmp` with one small synchronous hipMemcpy at call time (the
// values travel to the kernels as launch arguments either way).  Returns the pointer to read, or nullptr when the copy
// failed.  hipPointerGetAttributes fails for an ordinary host pointer (or says "unregistered"), which is the host case.
template <typename T>
inline const T *host_view(const T *p, T *tmp)
{
    if (!p) return nullptr;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError(); // not an error of ours: a plain host pointer
        return p;
    }
    if (attr.type != hipMemoryTypeDevice) return p; // host, registered host, managed: readable here
    if (hipMemcpy(tmp, p, sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return tmp;
}

} // namespace orbfe

struct orbfe_ctx {
    orbfe_config cfg;
    orbfe::DeviceGeom g;
    uint8_t *d_pyr = nullptr;       // [max_batch][frame_stride]
    uint8_t *d_pyr_alloc = nullptr; // the allocation: d_pyr lies kPyrGuardRows rows of level 0 inside it, and as many follow
                                    // the last frame, so that detect's tile loads need no range tests (batch_kernels.hip 4.2 A)
    uint32_t *d_cellkey = nullptr;  // [max_batch][K]
    uint4 *d_sel = nullptr;         // [max_batch][cap] selected keypoints in cell order: {cell, x | y << 16, key, 0}
    int32_t *d_selcount = nullptr;  // [max_batch]
    uint16_t *d_cellslot = nullptr; // [max_batch][K] record slot of each cell's keypoint, 0xFFFF = not selected
    uint4 *d_momw_tile = nullptr;   // describe (tile form): int8 weight fragments of the moment MFMAs
    // describe (tile form, degrees-as-radians regime): the rotated pattern as a table over the orientation
    // (steer_table.cpp): break points, per interval and lane the 8 sample offsets, the interval that holds 0
    float *d_steer_breaks = nullptr;
    uint4 *d_steer_table = nullptr;       // offsets for the tile kernel's LDS pitch
    uint4 *d_steer_table_patch = nullptr; // ... for the patch kernel's
    int n_steer_breaks = 0, steer_central = 0;
    uint64_t steer_sched_mask[4] = {0, 0, 0, 0};
    int describe_patch = 0;         // 1: sparse regime, patch kernel for large calls; 2 / -1: forced by ORBFE_DESCRIBE=patch / tile
    uint8_t *d_mdesc = nullptr;     // [max_batch][cap][32]  matcher scratch: dense descriptors
    uint8_t *d_mpos = nullptr;      // [max_batch][cap] float2 matcher scratch: positions
    int32_t *d_bend = nullptr;      // [max_batch][K]   windowed matcher: end of every cell bucket in d_bsorted
    uint16_t *d_bsorted = nullptr;  // [max_batch][cap] windowed matcher: record indices sorted by cell
    uint32_t *d_bd32 = nullptr;     // [max_batch][cap] reference-mode matcher: compressed 32-bit descriptors
    uint4 *d_mexp = nullptr;        // [max_batch][cap_pad][8]  MFMA matcher: descriptors as e2m1 fragments (cap <= 16384)
    float *d_mkey = nullptr;        // [max_batch][cap_pad]     MFMA matcher: -(popcount * 16384 + index)
    int detect_groups = 0;          // 0: by launch size; 1 / 2: forced by ORBFE_DETECT_GROUPS=single / multi (detect_tile_kernel)
    int match_form = 0;             // 0: by call size; 1 / 2: forced by ORBFE_MATCH=stream / tile (match_mfma.hip)
    int cap_pad = 0;                // cap rounded up to 16
    int cellkey_clean = 0;          // frames whose cell keys the last pyramid build left cleared (0 once detect ran)
    hipStream_t clean_stream = nullptr;      // ... on this stream
    unsigned long long clean_capture = 0;    // ... in this graph capture (0 = eager)
    uint4 *d_momw = nullptr;        // describe: int8 weight fragments of the moment MFMAs (make_moment_weights)
    orbfe::TileDesc *d_tiles = nullptr;
    int n_tiles = 0;
    char err[512] = {0};
};
