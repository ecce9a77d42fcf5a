"""round 5 diagnostic: time orbfe_match_batch alone (C2 shape: 4096 frames x 2000 records) for the matcher forms, with the
normal candidate stream and with every load hitting one block (max_distance < -1000: wrong results, L1-resident operands)."""
import os, sys, time
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
s = torch.cuda.current_stream().cuda_stream
for v2 in sys.argv[1:] or ["", "42", "41", "22", "21", "82"]:
    if v2: os.environ["ORBFE_MATCH_V2"] = v2
    else: os.environ.pop("ORBFE_MATCH_V2", None)
    for maxd in (256, -2000, -4000):
        for _ in range(5): ctx.match_batch(d_rec.data_ptr(), d_cnt.data_ptr(), B, 1, -1, maxd, d_idx.data_ptr(), None, s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ctx.match_batch(d_rec.data_ptr(), d_cnt.data_ptr(), B, 1, -1, maxd, d_idx.data_ptr(), None, s)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("V2=%-3s max_dist %5d: %.4f ms per %d pairs (x2 = per 4096-frame step: %.3f ms)" % (v2 or "-", maxd, ms, B - 1, 2 * ms), flush=True)
