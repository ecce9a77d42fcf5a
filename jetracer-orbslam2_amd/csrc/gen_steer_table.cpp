// gen_steer_table.cpp -- build-time generator of steer_events.inc (host program, run by the Makefile).
//
// rBRIEF samples the patch at RN(px * cos - py * sin), RN(px * sin + py * cos) for 512 pattern points (orb.cu:12-14,
// :42-46), where the steering angle is the stored orientation in RADIANS multiplied by pi / 180 (the reference's
// degrees-as-radians quirk Q7), i.e. |theta| <= 0.0549.  Over that range a rotated coordinate takes at most three
// values, p - 1, p, p + 1, and only coordinates with a partner |q| >= 10 ever leave p: as a function of the
// orientation (a float in [-pi, pi]) the whole rotated pattern is PIECEWISE CONSTANT with a few hundred pieces.
// The tile describe kernel therefore looks the 512 sample offsets up (one 16-byte load per lane) instead of
// computing them (80 vector instructions per keypoint): DESIGN.md 4.3.
//
// This program finds every orientation at which a coordinate changes, with the SAME float arithmetic the kernels and
// the oracle use (include/orbfe_math.h, no contraction): per coordinate and side, bisection on the ordered float
// domain brackets the change, then every float within +-kWindow ulps of it is evaluated, so that numerical flicker
// around the crossing (the computed value is not exactly monotone) is recorded change by change.  The argument that
// nothing changes outside those windows (the exact value is monotone in the angle with slope >= 0.28 / radian of
// theta, the computed one is within ~1e-6 of it, a window is >= 20x that) is backed by an EXHAUSTIVE check on the
// GPU: orbfe_selfcheck_steer_table compares the table with the arithmetic for every float in [-pi, pi]
// (tests/test_gpu_round3.py).
//
// Output: the break points (orientation bit patterns, ascending by value), per break point the coordinates that
// change and their new values, and the coordinates at -pi.  liborbfe expands that to LDS offsets per interval when a
// context is created (steer_table.cpp).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/orbfe_math.h"
#include "../../include/orbfe_pattern.h"

static const int8_t kPattern[ORBFE_PATTERN_TESTS * 4] = {ORBFE_PATTERN_VALUES};
static const int kWindow = 2048; // ulps scanned on either side of a bracketed change

// ordered key of a float: monotone in the value, -0 and +0 share key 0
static int64_t key_of(float f)
{
    uint32_t b;
    memcpy(&b, &f, 4);
    return (b & 0x80000000u) ? -(int64_t)(b & 0x7FFFFFFFu) : (int64_t)b;
}
static float float_of(int64_t k)
{
    const uint32_t b = k < 0 ? (0x80000000u | (uint32_t)(-k)) : (uint32_t)k;
    float f;
    memcpy(&f, &b, 4);
    return f;
}

// coordinate c = 2 * point + (0: row, 1: column), point = 2 * test + (0: P, 1: Q); exactly orb_describe_lds
static int coord_at(float angle, int c)
{
    const float ang = angle * ORBFE_DEG2RAD_F;
    float a, b;
    orbfe_sincosf(ang, &b, &a);
    const int point = c >> 1, test = point >> 1, which = point & 1;
    const float px = (float)kPattern[4 * test + 2 * which], py = (float)kPattern[4 * test + 2 * which + 1];
    if (c & 1) {
        const float p3 = px * a, p4 = py * b;
        return (int)__builtin_rintf(p3 - p4);
    }
    const float p1 = px * b, p2 = py * a;
    return (int)__builtin_rintf(p1 + p2);
}

struct Event {
    int64_t key; // first orientation (ordered key) at which the coordinate has the new value
    int coord, value;
};

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    const float pi = ORBFE_PI_F;
    const int64_t kmin = key_of(-pi), kmax = key_of(pi);
    std::vector<Event> events;
    std::vector<int> initial(1024);
    for (int c = 0; c < 1024; c++) {
        initial[c] = coord_at(-pi, c);
        const int v0 = coord_at(0.0f, c);
        const int vend[2] = {initial[c], coord_at(pi, c)};
        for (int side = 0; side < 2; side++) {
            if (vend[side] == v0) continue;
            int64_t lo = side ? 0 : kmin, hi = side ? kmax : 0; // g(lo) != g(hi)
            const int vlo = coord_at(float_of(lo), c);
            while (hi - lo > 1) {
                const int64_t mid = lo + (hi - lo) / 2;
                if (coord_at(float_of(mid), c) == vlo) lo = mid;
                else hi = mid;
            }
            int64_t k0 = std::max(kmin, hi - kWindow), k1 = std::min(kmax, hi + kWindow);
            int prev = coord_at(float_of(k0), c);
            for (int64_t k = k0 + 1; k <= k1; k++) {
                const int v = coord_at(float_of(k), c);
                if (v != prev) events.push_back(Event{k, c, v});
                prev = v;
            }
        }
    }
    std::sort(events.begin(), events.end(), [](const Event &x, const Event &y) { return x.key != y.key ? x.key < y.key : x.coord < y.coord; });
    std::vector<uint32_t> break_bits;
    std::vector<int> start;
    for (size_t i = 0; i < events.size(); i++) {
        if (i == 0 || events[i].key != events[i - 1].key) {
            const float f = float_of(events[i].key);
            uint32_t b;
            memcpy(&b, &f, 4);
            break_bits.push_back(b);
            start.push_back((int)i);
        }
    }
    start.push_back((int)events.size());
    FILE *o = fopen(argv[1], "w");
    if (!o) return 1;
    fprintf(o, "// generated by gen_steer_table.cpp (window %d ulps): %zu break points, %zu coordinate changes\n", kWindow,
            break_bits.size(), events.size());
    fprintf(o, "static const int kSteerBreaks = %zu, kSteerEvents = %zu;\n", break_bits.size(), events.size());
    fprintf(o, "static const uint32_t kSteerBreakBits[] = {");
    for (size_t i = 0; i < break_bits.size(); i++) fprintf(o, "%s0x%08Xu,", i % 8 ? " " : "\n    ", break_bits[i]);
    fprintf(o, "\n};\nstatic const uint32_t kSteerEventStart[] = {");
    for (size_t i = 0; i < start.size(); i++) fprintf(o, "%s%d,", i % 16 ? " " : "\n    ", start[i]);
    fprintf(o, "\n};\nstatic const uint16_t kSteerEventCoord[] = {");
    for (size_t i = 0; i < events.size(); i++) fprintf(o, "%s%d,", i % 16 ? " " : "\n    ", events[i].coord);
    fprintf(o, "\n};\nstatic const int8_t kSteerEventValue[] = {");
    for (size_t i = 0; i < events.size(); i++) fprintf(o, "%s%d,", i % 24 ? " " : "\n    ", events[i].value);
    fprintf(o, "\n};\nstatic const int8_t kSteerInitial[1024] = {");
    for (int c = 0; c < 1024; c++) fprintf(o, "%s%d,", c % 32 ? " " : "\n    ", initial[c]);
    fprintf(o, "\n};\n");
    fclose(o);
    fprintf(stderr, "gen_steer_table: %zu break points, %zu coordinate changes\n", break_bits.size(), events.size());
    return 0;
}
