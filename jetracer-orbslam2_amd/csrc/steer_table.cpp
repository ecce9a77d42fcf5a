// steer_table.cpp -- the rotated rBRIEF pattern as a table over the orientation (host side).
// steer_events.inc (generated at build time by gen_steer_table.cpp with the kernels' own float arithmetic) lists the
// orientations at which one of the 1024 rotated coordinates changes; this file expands it to what the tile describe
// kernel loads: per interval and lane the eight LDS byte offsets (relative to the keypoint, tile pitch `pitch`) of the
// lane's four tests, (P, Q) of rounds 0..3 as int16.  Constant data in, plain vectors out: no state.
#include <cstdint>
#include <cstring>
#include <vector>

#include "orbfe_internal.hpp"
#include "steer_events.inc"

namespace orbfe {

// Round schedule (tools/describe_schedule.py): in round r lane t gathers the samples of test 64 * p_t[r] + t, p_t a
// two-stage butterfly of four bits per lane (bit 0 swaps rounds 0,1; bit 1 rounds 2,3; then bit 2 rounds 0,2; bit 3
// rounds 1,3), chosen so that the 64 byte gathers of a round spread over the LDS banks.  Speed only: the kernel puts
// the ballots back in order with the four masks (any table gives the same descriptors; tools/experiments/
// profiling_probes.patch holds two other schedules the parity tests were run on).
// LDS cycles per keypoint for the 8 gathers: 52.3 plain -> 35.0 scheduled (ideal 16); tools/describe_schedule.py 40000 3
static const uint8_t kSchedBits[64] = {3, 14, 0, 12, 4, 8, 4, 3, 8, 15, 0, 2, 1, 12, 2, 15, 0, 8, 8, 8, 1, 7, 0, 1, 9, 12, 5, 5, 0, 4, 4, 2,
                                       1, 7, 10, 0, 0, 1, 5, 4, 2, 4, 4, 9, 0, 4, 1, 12, 4, 14, 1, 5, 0, 9, 12, 9, 2, 4, 0, 8, 4, 12, 0, 4};

static void sched_perm(int bt, int p[4])
{
    p[0] = 0, p[1] = 1, p[2] = 2, p[3] = 3;
    int t;
    if (bt & 1) { t = p[0]; p[0] = p[1]; p[1] = t; }
    if (bt & 2) { t = p[2]; p[2] = p[3]; p[3] = t; }
    if (bt & 4) { t = p[0]; p[0] = p[2]; p[2] = t; }
    if (bt & 8) { t = p[1]; p[1] = p[3]; p[3] = t; }
}

void build_steer_table(int pitch, std::vector<float> *breaks, std::vector<int16_t> *offsets, int *central, uint64_t sched_mask[4])
{
    for (int b = 0; b < 4; b++) {
        sched_mask[b] = 0;
        for (int l = 0; l < 64; l++)
            if (kSchedBits[l] >> b & 1) sched_mask[b] |= 1ull << l;
    }
    breaks->resize(kSteerBreaks);
    memcpy(breaks->data(), kSteerBreakBits, sizeof(float) * kSteerBreaks);
    int cur[1024];
    for (int c = 0; c < 1024; c++) cur[c] = kSteerInitial[c];
    offsets->assign((size_t)(kSteerBreaks + 1) * 512, 0);
    *central = 0;
    for (int iv = 0; iv <= kSteerBreaks; iv++) {
        if (iv > 0) {
            for (uint32_t e = kSteerEventStart[iv - 1]; e < kSteerEventStart[iv]; e++) cur[kSteerEventCoord[e]] = kSteerEventValue[e];
            if ((*breaks)[iv - 1] <= 0.0f) *central = iv; // interval index of an orientation = number of break points <= it
        }
        int16_t *o = offsets->data() + (size_t)iv * 512;
        for (int lane = 0; lane < 64; lane++) {
            int p[4];
            sched_perm(kSchedBits[lane], p);
            for (int r = 0; r < 4; r++)
                for (int which = 0; which < 2; which++) {
                    const int point = 2 * (64 * p[r] + lane) + which; // round r of this lane = test 64 p[r] + lane
                    o[lane * 8 + 2 * r + which] = (int16_t)(cur[2 * point] * pitch + cur[2 * point + 1]);
                }
        }
    }
}

} // namespace orbfe
