"""round 5 diagnostic: where a match_tile_kernel workgroup spends its time.  Needs the `stamps` build of tools/r5_tile_ablate_build.sh
(ORBFE_LIB=.../.variants/stamps/liborbfe.so): wave 0 of every workgroup leaves clock64() at kernel entry, loop start, loop end and exit in
the distance output.  Prints the mean prologue / loop / epilogue cycles and the loop's cycles per MFMA of one wave (16 = the pipe to itself,
32 = shared evenly by the two waves of a SIMD)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jetracer-orbslam2_amd"))
import numpy as np, torch, orbfe
B, n = 2048, 2000
rng = np.random.default_rng(1)
ctx = orbfe.Context(848, 480, levels=1, cell=8, min_arc=9, max_features=n, max_batch=B)
rec = np.zeros((64, n), dtype=orbfe.KEYPOINT_DTYPE)
rec["desc"] = rng.integers(0, 256, (64, n, 32), dtype=np.uint8)
rec["score"] = 50
d_rec = torch.from_numpy(np.tile(rec.view(np.uint8).reshape(64, -1), (B // 64, 1)).reshape(-1)).cuda()
d_cnt = torch.full((B,), n, dtype=torch.int32, device="cuda")
d_idx = torch.zeros((B - 1) * n, dtype=torch.int32, device="cuda")
d_dist = torch.zeros((B - 1) * n, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(3): ctx.match_batch(d_rec.data_ptr(), d_cnt.data_ptr(), B, 1, -1, 256, d_idx.data_ptr(), d_dist.data_ptr(), s)
torch.cuda.synchronize()
d = d_dist.cpu().numpy().reshape(B - 1, n)
st = np.stack([np.ascontiguousarray(d[:, t * 512:t * 512 + 12]).view(np.int64) for t in range(4)], 1).reshape(-1, 6)  # [workgroup][stamp]
pro, loop, epi = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2]
steps = (n + 63) // 64
print("workgroups %d; clock64 cycles, mean (p10 / p90):" % len(st))
for name, v in (("prologue", pro), ("loop", loop), ("epilogue", epi)):
    print("  %-9s %9.0f (%7.0f / %7.0f)" % (name, v.mean(), np.percentile(v, 10), np.percentile(v, 90)))
print("  loop: %.1f clock64 ticks per step, %.2f per MFMA of one wave (%d steps x 64)" % (loop.mean() / steps, loop.mean() / steps / 64, steps))
life, wall = st[:, 3] - st[:, 0], st[:, 5] - st[:, 4]  # core-clock ticks, 100 MHz ticks
print("  core clock while the kernel runs: %.0f MHz (clock64 / wall_clock64 over a workgroup's lifetime)" % (life.sum() / wall.sum() * 100.0))
span = st[:, 5].max() - st[:, 4].min()
print("  kernel span %.1f us; workgroup slots in use: sum of lifetimes / (span x 512 slots) = %.3f" % (span / 100.0, wall.sum() / (span * 512.0)))
print("  lifetime %.0f ticks = %.1f us; 2 x 2048 MFMAs x 16 cycles = 65 536 ticks per pair of workgroups on a SIMD" % (life.mean(), wall.mean() / 100.0))
