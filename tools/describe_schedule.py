#!/usr/bin/env python3
"""Round schedule of the tile describe kernel's 256 rBRIEF tests.

In round r (0..3) lane t of the wave gathers the two samples of ONE test and the 64-bit ballot of the outcomes is a
descriptor word.  With the plain schedule (round r = tests 64 r .. 64 r + 63) each of the 8 byte gathers per keypoint
hits the 32 LDS banks at random: ~6.5 LDS cycles per ds_read_u8 instead of 2, and since the sample offsets come from
the orientation table the gathers are what the loop waits for.  Lane t may take its four tests (t, 64 + t, 128 + t,
192 + t) in any order without moving a result bit to another lane: round r then holds bit t of word p_t[r], and the
words are restored by masked swaps of whole ballots on the scalar unit.  This tool searches the per-lane orders
reachable by a two-stage butterfly (bit 0: swap rounds 0,1; bit 1: rounds 2,3; then bit 2: rounds 0,2; bit 3: rounds
1,3 -- four 64-bit masks, 16 scalar instructions per keypoint) for the fewest bank-conflict cycles over the small
steering angles of the degrees-as-radians regime and the four byte alignments of a keypoint, for the kernel's LDS pitch.

Prints the table for csrc/steer_table.cpp (kSchedBits).  The schedule changes speed only: any table gives the same
descriptors.   usage: tools/describe_schedule.py [iterations] [seed]
"""
import os
import random
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PITCH = 112  # TileGeom<15>::kPitch


def pattern():
    src = open(os.path.join(ROOT, "include", "orbfe_pattern.h")).read()
    m = re.search(r"#define ORBFE_PATTERN_VALUES(.*?)\n\n", src, re.S)
    vals = [int(v) for v in re.findall(r"-?\d+", m.group(1))]
    return np.array(vals[:1024]).reshape(256, 4)


def perm_of(bt):
    p = [0, 1, 2, 3]  # p[r] = descriptor word evaluated in round r
    if bt & 1:
        p[0], p[1] = p[1], p[0]
    if bt & 2:
        p[2], p[3] = p[3], p[2]
    if bt & 4:
        p[0], p[2] = p[2], p[0]
    if bt & 8:
        p[1], p[3] = p[3], p[1]
    return p


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    pat = pattern()
    # 70 % of the orientations fall into the table's central piece (no coordinate moved), the rest spread out
    scen = [(t, x) for t in (0.0, 0.0, 0.0, 0.0, 0.02, -0.02, 0.045, -0.045) for x in (40, 41, 42, 43)]
    dw = np.zeros((len(scen), 256, 2), dtype=np.int64)  # LDS dword of sample P / Q of every test
    for si, (th, x) in enumerate(scen):
        a, b = np.float32(np.cos(th)), np.float32(np.sin(th))
        for k in (0, 1):
            px, py = pat[:, 2 * k].astype(np.float32), pat[:, 2 * k + 1].astype(np.float32)
            col, row = np.rint(px * a - py * b).astype(int), np.rint(px * b + py * a).astype(int)
            dw[si, :, k] = ((40 + row) * PITCH + x + col) // 4

    def round_cost(tests):  # LDS cycles of the round's two ds_read_u8: 32-lane halves, 32 banks, same dword broadcasts
        tot = 0
        for si in range(len(scen)):
            for k in (0, 1):
                d = dw[si, tests, k]
                for g in (0, 1):
                    tot += np.bincount(np.unique(d[32 * g:32 * g + 32]) % 32, minlength=32).max()
        return tot

    bits = np.zeros(64, dtype=int)
    lanes = np.arange(64)
    assign = np.array([64 * r + lanes for r in range(4)])
    costs = [round_cost(assign[r]) for r in range(4)]
    best = sum(costs)
    start = best / len(scen)
    rng = random.Random(seed)
    for it in range(iters):
        ln, b = rng.randrange(64), 1 << rng.randrange(4)
        bits[ln] ^= b
        p = perm_of(bits[ln])
        new = assign.copy()
        for r in range(4):
            new[r, ln] = 64 * p[r] + ln
        nc = list(costs)
        for r in range(4):
            if new[r, ln] != assign[r, ln]:
                nc[r] = round_cost(new[r])
        if sum(nc) <= best:
            best, costs, assign = sum(nc), nc, new
        else:
            bits[ln] ^= b
    print("// LDS cycles per keypoint for the 8 gathers: %.1f plain -> %.1f scheduled (ideal 16); tools/describe_schedule.py %d %d"
          % (start, best / len(scen), iters, seed))
    print("static const uint8_t kSchedBits[64] = {" + ", ".join(str(int(v)) for v in bits) + "};")


if __name__ == "__main__":
    main()
