#!/bin/bash
# round 5 diagnostic build (.variants/phstamps): detect_tile_kernel and describe_tile_kernel add up, per workgroup, the s_memtime ticks
# wave 0 spends between its barriers (= the kernel's phases) into a device array; orbfe_debug_phase() (this build only) reads it.
# tools/r5_phase_stamps.py runs bench steps on it.  Scripted edits of a scratch copy, like tools/r5_tile_ablate_build.sh.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
W=$(mktemp -d /tmp/orbfe_v.XXXX); mkdir -p $W/jetracer-orbslam2_amd; cp -r include $W/; cp -r jetracer-orbslam2_amd/csrc $W/jetracer-orbslam2_amd/; rm -rf $W/jetracer-orbslam2_amd/csrc/.obj
python3 - "$W/jetracer-orbslam2_amd/csrc/batch_kernels.hip" <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
def once(old, new):
    global s
    assert s.count(old) == 1, old
    s = s.replace(old, new)
once("constexpr int kSteerMaxBreaks = 224;", "__device__ unsigned long long g_phase[4096 * 16];\n#define STAMP(i, a, b) do { if (threadIdx.x == 0) atomicAdd(&g_phase[(((blockIdx.x + 977u * blockIdx.y) & 4095u) << 4) + ((i) & 15) ], (unsigned long long)((b) - (a))); } while (0)\nconstexpr int kSteerMaxBreaks = 224;")
# ---- detect: A = tile load, B = compass + compaction, C = ring test, D = 3x3 maximum + cell keys
once("    const int l = td.level;\n    const int W = g.lv[l].w, H = g.lv[l].h, P = g.lv[l].pitch;\n    const uint8_t *img = pyr + (size_t)f * g.frame_stride + g.lv[l].offset;\n    const int x0 = td.tx * kTileW",
     "    const long long tq0 = clock64(), wq0 = wall_clock64();\n    const int l = td.level;\n    const int W = g.lv[l].w, H = g.lv[l].h, P = g.lv[l].pitch;\n    const uint8_t *img = pyr + (size_t)f * g.frame_stride + g.lv[l].offset;\n    const int x0 = td.tx * kTileW")
once("    if (tid == 0) s_qcount = 0;\n    __syncthreads();\n", "    if (tid == 0) s_qcount = 0;\n    __syncthreads();\n    const long long tq1 = clock64();\n")
once("    __syncthreads(); // the queue is complete\n", "    __syncthreads(); // the queue is complete\n    const long long tq2 = clock64();\n")
once("    __syncthreads(); // every wave's scores are in s_sc\n", "    __syncthreads(); // every wave's scores are in s_sc\n    const long long tq3 = clock64();\n")
once("    if (!lds_cells) return;\n    __syncthreads();\n    for (int i = tid; i < ncx * ncy; i += 256) {\n        const uint32_t key = s_key[i];\n        if (key == 0u) continue;\n        const int cx = (x0 >> lc) + (i & (ncx - 1)), cy = (y0 >> lc) + (i >> lnc);\n        if (cx < g.cells_x && cy < g.cells_y)\n            atomicMax(&cellkey[(size_t)f * g.K + cy * g.cells_x + cx], key);\n    }\n",
     "    if (lds_cells) {\n    __syncthreads();\n    for (int i = tid; i < ncx * ncy; i += 256) {\n        const uint32_t key = s_key[i];\n        if (key == 0u) continue;\n        const int cx = (x0 >> lc) + (i & (ncx - 1)), cy = (y0 >> lc) + (i >> lnc);\n        if (cx < g.cells_x && cy < g.cells_y)\n            atomicMax(&cellkey[(size_t)f * g.K + cy * g.cells_x + cx], key);\n    }\n    }\n    { const long long tq4 = clock64(); STAMP(0, tq0, tq1); STAMP(1, tq1, tq2); STAMP(2, tq2, tq3); STAMP(3, tq3, tq4); STAMP(4, 0, 1); STAMP(5, wq0, wall_clock64()); STAMP(6, 0, nq); STAMP(7, tq0, tq4); }\n")
# ---- describe (tile form, first pass): stage = tile DMA + keypoint list, A = moments, B = angles, C = descriptors
once("    const int W = g.lv[l].w, H = g.lv[l].h; // the sampled level (l = 0 unless DL)\n", "    const long long tq0 = clock64(), wq0 = wall_clock64();\n    const int W = g.lv[l].w, H = g.lv[l].h; // the sampled level (l = 0 unless DL)\n")
once("        const int nkp = s_nkp;\n        if (DL) cursor = s_cursor;\n", "        const int nkp = s_nkp;\n        const long long tq1 = clock64();\n        if (DL) cursor = s_cursor;\n")
once("        __syncthreads();\n        // ---- phase B: one lane per keypoint", "        __syncthreads();\n        const long long tq2 = clock64();\n        // ---- phase B: one lane per keypoint")
once("        __syncthreads();\n        // ---- phase C: descriptors, keypoints dealt round-robin", "        __syncthreads();\n        const long long tq3 = clock64();\n        // ---- phase C: descriptors, keypoints dealt round-robin")
once("        if (!DL || cursor >= ngroups) break;\n", "        { const long long tq4 = clock64(); STAMP(8, tq0, tq1); STAMP(9, tq1, tq2); STAMP(10, tq2, tq3); STAMP(11, tq3, tq4); STAMP(12, 0, 1); STAMP(13, wq0, wall_clock64()); STAMP(14, 0, nkp); STAMP(15, tq0, tq4); }\n        if (!DL || cursor >= ngroups) break;\n")
once("} // extern \"C\"", "int orbfe_debug_phase(unsigned long long *out, int reset)\n{\n    static unsigned long long z[4096 * 16];\n    if (hipMemcpyFromSymbol(z, HIP_SYMBOL(orbfe::g_phase), sizeof z) != hipSuccess) return -1;\n    for (int i = 0; i < 32; i++) out[i] = 0;\n    for (int i = 0; i < 4096 * 16; i++) out[i & 15] += z[i];\n    if (reset) { for (auto &x : z) x = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(orbfe::g_phase), z, sizeof z) != hipSuccess) return -1; }\n    return 0;\n}\n} // extern \"C\"")
open(p, "w").write(s)
PY
OUT=$ROOT/jetracer-orbslam2_amd/.variants/phstamps; mkdir -p $OUT
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function"
O=jetracer-orbslam2_amd/csrc/.obj
/opt/rocm/bin/hipcc $F -I$O -c -o $OUT/batch.o $W/jetracer-orbslam2_amd/csrc/batch_kernels.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/liborbfe.so $O/stage_kernels.o $OUT/batch.o $O/match_mfma.o $O/align_depth.o $O/ingest.o $O/wire_bson.o $O/pose_host.o $O/steer_table.o
rm -rf $W $OUT/batch.o; echo built $OUT/liborbfe.so
