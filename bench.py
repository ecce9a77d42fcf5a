#!/usr/bin/env python3
"""bench.py -- ORB front-end throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode c2|ref|c3|c4|c5|match|align] [--scene dense|survey]

One "step" = one pass of the hot path over one batch of synthetic frames already resident in HBM:
orbfe_extract (blur + pyramid -> fused FAST/NMS -> selection -> orientation + rBRIEF -> 52-byte
records) followed by the Hamming matcher.  Rank 0 prints ONE JSON line.

Modes (BASELINE.json configs):
  c2   configs[1], the metric: 640x480 mono, 8-level pyramid, 2000 features/frame; here cell 8
       (4800 cells, detection on levels 0..3), FAST-9 t = 13, top-2000 by (score desc, cell asc),
       256-bit brute-force matching t-1 -> t; 4096 frames per GPU per step (weak scaling; 5 GB of
       inputs, pyramids, records and matcher scratch resident in HBM, far beyond the 256 MB Infinity
       Cache, so no step finds its data cached by the previous one, and 20 steps are > 100 ms of GPU
       work; --batch 256 is round 1-2's size).
  ref  the reference-parity configuration (cell 32, FAST-12, 6 levels, <= 300 keypoints, 32-bit
       windowed matcher) -- a parity case, not the metric.
  c3   configs[2]: RealSense-shaped stereo 848x480 pairs (right = left shifted 3 px + noise),
       extract both + windowed 256-bit match left -> right; 128 pairs per GPU per step.
  c4   configs[3]: 64 frames 1280x720 in total, sharded over the ranks (strong scaling), records
       gathered on rank 0.
  c5   configs[4]: ONE 3840x2160 frame, 12 levels, 8000 features; detection tiles sharded over the
       ranks, per-cell keys merged by all-reduce(MAX), then selection + description.
  match  SURVEY.md 8d's matcher microbench: nA = nB in {405, 2000, 8192}, random 256-bit descriptors
       (C3's reference shape, C2/C3's budget, C5's), brute force 256-bit and the reference's 32-bit /
       +-2 px window form; one JSON line with Gpairs/s and the MFMA fraction per size.  1 GPU.

  align  SURVEY.md 8f-2's producing half: orbfe_align_depth_batch (depth -> colour camera, cuda-align.cu:366-399) over
       1024 RealSense-shaped 848x480 depth frames per call, D435-like rig; HBM roofline on 6 B per pixel.  1 GPU.

Multi-GPU: one process per GPU.  Under torchrun (WORLD_SIZE set) this process is one rank; with
--gpus N > 1 and no WORLD_SIZE the parent starts N child ranks itself BEFORE touching the GPU and
relays rank 0's line.  The timed region at N > 1 ends when the keypoint gather is complete on
rank 0: records and counts travel through liborbfe_dist.so (C++ host code on RCCL:
grouped ncclSend / ncclRecv on its own stream, overlapped with the next step's kernels).  The
rate without the gather is reported beside it as `no_gather`.

Only the cpu_baseline leg imports oracle/ (the CPU restatement, timed as a baseline).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "jetracer-orbslam2_amd"))

EXT = dict(levels=8, cell=8, min_arc=9, max_features=2000)
MODES = {
    "c2": dict(width=640, height=480, cfg=EXT, match=dict(mode=1, window=-1, max_distance=256), batch=4096, stride=1,
               scaling="weak",
               workload="640x480 mono, 8-level pyramid, 2000 features/frame (cell 8, FAST-9 t=13, top-2000), "
                        "256-bit brute-force match t-1->t"),
    "ref": dict(width=640, height=480, cfg=dict(levels=6, cell=32, min_arc=12, max_features=0),
                match=dict(mode=0, window=2, max_distance=4), batch=256, stride=1, scaling="weak",
                workload="640x480 mono, reference-parity mode (6 levels, cell 32, FAST-12 t=13, <=300 keypoints), "
                         "32-bit windowed match t-1->t"),
    "c3": dict(width=848, height=480, cfg=EXT, match=dict(mode=1, window=16, max_distance=64), batch=256, stride=2,
               scaling="weak",
               workload="848x480 stereo pairs (right = left shifted 3 px, +-2 noise), 8 levels, 2000 features/frame, "
                        "256-bit match left->right within +-16 px, distance <= 64"),
    "c4": dict(width=1280, height=720, cfg=EXT, match=dict(mode=1, window=-1, max_distance=256), batch=64, stride=1,
               scaling="strong",
               workload="64 frames 1280x720 in total sharded over the ranks, 8 levels, 2000 features/frame, 256-bit "
                        "brute-force match t-1->t, keypoint records gathered on rank 0"),
    "c5": dict(width=3840, height=2160, cfg=dict(levels=12, cell=16, min_arc=9, max_features=8000),
               match=None, batch=1, stride=1, scaling="strong",
               workload="one 3840x2160 frame, 12-level pyramid, 8000 features (cell 16, FAST-9, top-8000); detection "
                        "tiles sharded over the ranks, cell keys merged by all-reduce(MAX), then describe"),
}
FP4_MFMA_PEAK_TFLOPS = 10000.0
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# vector ALU: 256 CUs x 4 SIMDs x 32 lanes per clock at 2.4 GHz (the chip's 157.3 TFLOP/s fp32 / 2);
# full-rate instructions measure 58-64 T lane-ops/s under load (tools/valu_rate*.hip)
VALU_PEAK_TLANEOPS = 78.6


# --------------------------------------------------------------------------------------------
# launching N ranks from one command line (no torch / GPU call may precede this)
# --------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def child_commands(args, argv):
    """The N child (argv, env-additions) this parent starts for --gpus N."""
    port = int(os.environ.get("MASTER_PORT", "0")) or _free_port()
    out = []
    for r in range(args.gpus):
        env = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0",
               "ORBFE_BENCH_CHILD": "1"}
        out.append(([sys.executable, os.path.abspath(__file__)] + [a for a in argv if a != "--dry-run"], env))
    return out


def spawn_ranks(args, argv):
    cmds = child_commands(args, argv)
    if args.dry_run:
        print(json.dumps([{"argv": c, "env": e} for c, e in cmds]))
        return 0
    procs = []
    for r, (cmd, env) in enumerate(cmds):
        e = dict(os.environ)
        e.update(env)
        # rank 0 owns stdout (the JSON line); the other ranks' stdout goes to stderr
        procs.append(subprocess.Popen(cmd, env=e, stdout=None if r == 0 else sys.stderr))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


# --------------------------------------------------------------------------------------------
def algorithmic_bytes(width, height, levels, detect_levels, cells, k_out):
    """B_frame = P0 + 2 * sum(P_l) + 52 * K_out, and its per-kernel split."""
    p = [(width >> l) * (height >> l) for l in range(levels)]
    pyramid = p[0] + sum(p)                       # read input, write every level once
    detect = sum(p[:detect_levels]) + 4 * cells   # read detect levels once, write cell keys
    describe = 52 * k_out                         # write records (patch gathers hit L2/MALL)
    frame = p[0] + 2 * sum(p) + 52 * k_out
    return dict(frame=frame, pyramid=pyramid, detect=detect, describe=describe)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_share():
    """(threads to use, affinity count, cgroup CPU quota or None) of this process: the affinity mask, cut to the
    cgroup's cpu.max quota when there is one (a 1-GPU box is a 16-CPU share of a larger host)."""
    n = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(float(q) / float(per) + 0.5))
    except (OSError, ValueError):
        pass
    return (min(n, quota) if quota else n), n, quota


def cpu_baseline(frames, mode, seconds_1t=7.0, seconds_mt=10.0):
    """Oracle (CPU port of the reference semantics) timed on this host: extract + match, first on
    one thread, then frame-parallel on the box's CPU share."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    from concurrent.futures import ThreadPoolExecutor
    m = MODES[mode]
    ocfg = oracle.make_config(m["width"], m["height"], **m["cfg"])
    cores, affinity, quota = cpu_share()  # the threads actually used = what is reported as `cores`
    mm = m["match"]

    def comp(d):
        return ((d == 1).astype(np.uint32) << np.arange(32, dtype=np.uint32)).sum(1).astype(np.uint32)

    def extract(i):
        return oracle.extract_frame(frames[i % len(frames)], ocfg)["records"]

    def match(pair):
        a, b = pair
        pa, pb = np.stack([a["x"], a["y"]], 1), np.stack([b["x"], b["y"]], 1)
        if mm["mode"] == 1:
            oracle.match256(a["desc"], b["desc"], pa, pb, mm["window"], mm["max_distance"])
        else:
            oracle.match_keypoints(pa, comp(a["desc"]), pb, comp(b["desc"]), mm["window"], mm["max_distance"])

    stride = m["stride"]

    def run(threads, seconds):
        # time-bounded: chunks of `threads` frames (extract, then match the mode's pairs) until the
        # budget is spent, so the sample stays ~`seconds` whatever the host's real CPU share is
        chunk = threads + (threads % 2 if stride == 2 else 0)
        recs_prev, n_frames, n_pairs, kp = None, 0, 0, 0
        t0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:  # ctypes releases the GIL inside the oracle
            while time.perf_counter() - t0 < seconds:
                recs = list(ex.map(extract, range(n_frames, n_frames + chunk)))
                if mm is not None:
                    if stride == 2:  # stereo: (left, right) pairs only
                        prs = list(zip(recs[0::2], recs[1::2]))
                    else:            # temporal: t-1 -> t, chained across chunks
                        chain = ([recs_prev] if recs_prev is not None else []) + recs
                        prs = list(zip(chain[:-1], chain[1:]))
                    list(ex.map(match, prs))
                    n_pairs += len(prs)
                n_frames += len(recs)
                kp += sum(len(r) for r in recs)
                recs_prev = recs[-1]
        dt = time.perf_counter() - t0
        return kp / dt, n_frames / dt, n_frames, n_pairs, dt

    v1, f1, n1, p1, t1 = run(1, seconds_1t)
    vm, fm, nm, pm, tm = run(cores, seconds_mt)
    return dict(value=vm, unit="keypoints/s", cores=cores, affinity_cpus=affinity, cgroup_cpu_quota=quota, kind="port", frames_per_s=fm,
                single_thread=dict(value=v1, unit="keypoints/s", cores=1, frames_per_s=f1,
                                   sample="%d frames + %d pairs in %.1f s" % (n1, p1, t1)),
                cpu_model=cpu_model(),
                sample="%d frames extracted + %d frame pairs matched by the CPU oracle on %d threads in %.1f s "
                       "(and %d frames + %d pairs on 1 thread in %.1f s)" % (nm, pm, cores, tm, n1, p1, t1))


def make_scenes(synth, mode, scene, n_distinct, first_index):
    """Synthetic frames [n, H, W] u8.  `dense`: corner-rich scenes that fill the feature budget
    (rectangles of 6..32 px, 800 per 640x480 of area); `survey`: SURVEY.md 8d's generator (96
    rectangles up to a fifth of the frame, +-3 noise)."""
    import numpy as np
    m = MODES[mode]
    w, h = m["width"], m["height"]
    kw = dict(n_rects=96) if scene == "survey" else \
        dict(n_rects=800 * (w * h) // (640 * 480), min_size=6, max_size=32)
    if mode == "c3":  # (left, right) pairs: the same scene shifted by 3 px with fresh +-2 noise
        out = []
        for i in range(n_distinct // 2):
            a, b = synth.shifted_pair(w, h, first_index + i, dx=3, dy=0, **kw)
            out += [a, b]
        return np.stack(out)
    return synth.frames(w, h, n_distinct, first_index=first_index, kind="rects", **kw)


MATCH_SIZES = [  # nA = nB in {405, 2000, 8192}; a context whose record capacity is exactly n
    dict(n=405, width=848, height=480, cfg=dict(levels=1, cell=32, min_arc=12, max_features=0), frames=256,
         shape="C3 reference regime: 848x480, one keypoint per 32-px cell"),
    dict(n=2000, width=848, height=480, cfg=dict(levels=1, cell=8, min_arc=9, max_features=2000), frames=256,
         shape="C2 / C3 feature budget"),
    dict(n=8192, width=3840, height=2160, cfg=dict(levels=1, cell=16, min_arc=9, max_features=8192), frames=16,
         shape="C5: one 3840x2160 frame's budget"),
]


def match_microbench(args, torch, np, orbfe, dev, json_out):
    """--mode match: the matcher alone on random 256-bit descriptors (post_processing.cu:92-200 is the reference
    matcher).  Records are synthetic: keypoint i of every frame sits at the centre of cell i (cell order, as
    extraction emits them), odd frames shifted by one pixel, descriptors i.i.d. random; all n records valid."""
    s = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(20261004)
    out_sizes = []
    for S in MATCH_SIZES:
        n, w, h, B = S["n"], S["width"], S["height"], S["frames"]
        ctx = orbfe.Context(w, h, max_batch=B, **S["cfg"])
        assert ctx.cap == n, (ctx.cap, n)
        cell = S["cfg"]["cell"]
        cells_x = (w + cell - 1) // cell
        k = np.sort(rng.choice(ctx.K, n, replace=False)) if n < ctx.K else np.arange(n)
        rec = np.zeros((B, n), dtype=orbfe.KEYPOINT_DTYPE)
        rec["x"] = np.minimum((k % cells_x) * cell + cell // 2, w - 1)[None, :] + (np.arange(B) % 2)[:, None]
        rec["y"] = np.minimum((k // cells_x) * cell + cell // 2, h - 1)[None, :]
        rec["score"], rec["level"] = 100.0, 0
        rec["desc"] = rng.integers(0, 256, (B, n, 32), dtype=np.uint8)
        d_rec = torch.from_numpy(rec.view(np.uint8).reshape(-1)).to(dev)
        d_cnt = torch.full((B,), n, dtype=torch.int32, device=dev)
        d_idx = torch.zeros((B - 1) * n, dtype=torch.int32, device=dev)
        d_dst = torch.zeros((B - 1) * n, dtype=torch.int32, device=dev)
        pairs = (B - 1) * n * n
        entry = {"n": n, "frames_per_call": B, "frame_pairs_per_call": B - 1, "descriptor_pairs_per_call": pairs,
                 "shape": S["shape"], "algorithmic_bytes_per_call": {}}
        for label, mode, window, maxd, per in (("brute_force_256bit", 1, -1, 256, 40), ("reference_32bit_window2", 0, 2, 4, 12)):
            def call():
                ctx.match_batch(d_rec.data_ptr(), d_cnt.data_ptr(), B, mode, window, maxd, d_idx.data_ptr(), d_dst.data_ptr(), s)
            for _ in range(10):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            call()
            e1.record()
            torch.cuda.synchronize()
            reps = max(10, min(2000, int(100.0 / max(e0.elapsed_time(e1), 1e-3))))  # ~100 ms of calls
            e0.record()
            for _ in range(reps):
                call()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            disp = ctx.dispatch_info(B, mode, window)
            ab = (per * 2 * n + 8 * n) * (B - 1)  # B_match = per * (nA + nB) + 8 * nA per frame pair
            r = {"ms_per_call": ms, "calls_timed": reps, "kernels": disp["match"], "examines": disp["match_examines"],
                 "algorithmic_bytes_per_call": ab, "hbm_frac_of_8TBps": ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                 "matched": int((d_idx[:n] >= 0).sum().item())}
            if disp["match_examines"] == "all_pairs":
                r["gpairs_per_s"] = pairs / (ms * 1e-3) / 1e9
                if "mfma" in disp["match"] or "match_tile_kernel" in disp["match"]:
                    r["mfma"] = {"bound": "mfma", "unit": "TFLOP/s", "peak": FP4_MFMA_PEAK_TFLOPS,
                                 "achieved": 512.0 * pairs / (ms * 1e-3) / 1e12,
                                 "frac": 512.0 * pairs / (ms * 1e-3) / 1e12 / FP4_MFMA_PEAK_TFLOPS}
            else:  # the cell index looks only at the candidates inside the window: nA * nB is what the REFERENCE would visit
                r["nominal_gpairs_per_s"] = pairs / (ms * 1e-3) / 1e9
                r["note"] = ("the reference visits all nA x nB pairs (post_processing.cu:134-170); the cell-indexed kernel "
                             "examines only the window's cells, so this is an equivalent rate, not comparisons done")
            entry[label] = r
        out_sizes.append(entry)
        ctx.close()
        del d_rec, d_idx, d_dst
    head = next(e for e in out_sizes if e["n"] == 2000)["brute_force_256bit"]
    out = {"metric": "matcher Gpairs/sec, brute-force 256-bit Hamming, nA = nB = 2000 (SURVEY.md 8d microbench)",
           "value": head["gpairs_per_s"], "unit": "Gpairs/s", "n_gpus": 1, "steps": head["calls_timed"], "warmup": 10,
           "ms_per_step": head["ms_per_call"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u8 (exact Hamming distance as an e2m1 matrix product, f32 accumulate)", "data": "synthetic",
           "config": {"workload": "matcher microbench: nA = nB in {405, 2000, 8192}, random 256-bit descriptors, "
                                  "brute force 256-bit and reference 32-bit / +-2 px window", "mode": "match"},
           "roofline": dict(head["mfma"], kernel=head["kernels"], note="the matcher's roof is the matrix cores (dense FP4 "
                            "MFMA peak ~10 PFLOP/s), its HBM traffic is negligible by construction (SURVEY.md 8d)",
                            hbm={"bound": "hbm", "frac": head["hbm_frac_of_8TBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                 "achieved": head["hbm_frac_of_8TBps"] * HBM_PEAK_GBPS, "traffic": None}),
           "sizes": out_sizes, "cpu_baseline": None}
    if not args.no_cpu_baseline:  # the oracle's brute-force matcher on one thread, a bounded sample
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle
        a = rng.integers(0, 256, (2000, 32), dtype=np.uint8)
        b = rng.integers(0, 256, (2000, 32), dtype=np.uint8)
        t0, calls = time.perf_counter(), 0
        while time.perf_counter() - t0 < 5.0:
            oracle.match256(a, b)
            calls += 1
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": calls * 4e6 / dt / 1e9, "unit": "Gpairs/s", "cores": 1, "kind": "port",
                               "cpu_model": cpu_model(), "sample": "%d calls of oracle_match256 (2000 x 2000) in %.1f s" % (calls, dt)}
    json_out.write(json.dumps(out) + "\n")
    json_out.flush()


def align_bench(args, torch, np, orbfe, dev, json_out):
    """--mode align: SURVEY.md 8f-2's producing half, orbfe_align_depth_batch (cuda-align.cu:366-399), on RealSense-shaped
    848x480 depth frames with a D435-like rig (synth.rig('d435')).  A step = one call over `batch` frames resident in HBM.
    Algorithmic bytes: 2 B read + 4 B written per pixel; the kernel is HBM-bound in principle."""
    import ctypes as C
    from orbfe import synth
    w, h = 848, 480
    B = args.batch or 1024
    n_distinct = max(1, min(args.distinct, B, 32))
    base = synth.depth_frames(w, h, n_distinct, first_index=500)
    d, o, e, scale = synth.rig("d435", w, h)
    mk = lambda t: orbfe.Intrinsics(t[0], t[1], t[2], t[3], t[4], t[5], t[6], (C.c_float * 5)(*t[7]))
    di, oi = mk(d), mk(o)
    ex = orbfe.Extrinsics((C.c_float * 9)(*e[0]), (C.c_float * 3)(*e[1]))
    src = torch.from_numpy(base.view(np.int16)).to(dev)[torch.arange(B, device=dev) % n_distinct].contiguous()
    out = torch.zeros((B, h * w), dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream

    def call():
        orbfe.check(orbfe.lib().orbfe_align_depth_batch(out.data_ptr(), w * h, src.data_ptr(), w * h, B, scale, C.byref(di),
                                                        C.byref(oi), C.byref(ex), s))
    for _ in range(max(args.warmup, 3)):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        call()
    e1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms = e0.elapsed_time(e1) / args.steps
    ab = 6.0 * w * h * B
    covered = float((out[:n_distinct] != 0).float().mean().item())
    prof = {}
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("align", {})
    except Exception:
        prof = {}
    stale = not prof or prof.get("csrc_sha256") != orbfe.source_hash()
    traffic = None if stale else prof.get("align", 0.0) * B / float(prof.get("batch", B))
    res = {"metric": "depth frames aligned to the colour camera per second (align_depth_to_other), 848x480",
           "value": B * args.steps / elapsed, "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": max(args.warmup, 3),
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32 (projection arithmetic) / u16 -> u32 (depth)", "data": "synthetic",
           "config": {"workload": "align_depth_to_other over %d depth frames 848x480 per call, D435-like rig (depth 87 deg, colour 69 deg, "
                                  "15 mm baseline), 15 %% holes" % B, "mode": "align", "frames_per_gpu_per_step": B,
                      "distinct_frames_per_rank": n_distinct, "frames_per_launch": int(os.environ.get("ORBFE_ALIGN_CHUNK", "128")),
                      "output_pixels_covered": covered},
           "pixels_per_s": B * w * h * args.steps / elapsed,
           "roofline": {"bound": "hbm", "achieved": ab / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": traffic,
                        "kernel": "align_fill_kernel + align_splat_kernel (one call = ceil(frames / frames_per_launch) launches of each)",
                        "algorithmic_bytes_per_launch": ab, "avg_launch_ms": ms,
                        "note": "algorithmic bytes = 2 B read + 4 B written per pixel (VERDICT r3 item 3); time = HIP events around "
                                "the timed calls on the work stream; `launch` here = one orbfe_align_depth_batch call",
                        "pmc": {"stale": stale}},
           "cpu_baseline": None}
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle
        from concurrent.futures import ThreadPoolExecutor
        cores, affinity, quota = cpu_share()
        omk = lambda t: oracle.Intrinsics(t[0], t[1], t[2], t[3], t[4], t[5], t[6], (C.c_float * 5)(*t[7]))
        odi, ooi = omk(d), omk(o)
        oex = oracle.Extrinsics((C.c_float * 9)(*e[0]), (C.c_float * 3)(*e[1]))

        def one(i):
            oracle.align_depth_to_other(base[i % n_distinct], scale, w, h, odi, ooi, oex)

        def run(threads, seconds):
            n, t0 = 0, time.perf_counter()
            with ThreadPoolExecutor(threads) as tp:
                while time.perf_counter() - t0 < seconds:
                    list(tp.map(one, range(n, n + 4 * threads)))
                    n += 4 * threads
            return n / (time.perf_counter() - t0), n, time.perf_counter() - t0
        v1, n1, t1 = run(1, 4.0)
        vm, nm, tm = run(cores, 8.0)
        res["cpu_baseline"] = {"value": vm, "unit": "frames/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
                               "single_thread": {"value": v1, "unit": "frames/s", "cores": 1},
                               "sample": "%d frames by the CPU oracle (the reference's four launches restated, int2 map included) on %d "
                                         "threads in %.1f s, %d on 1 thread in %.1f s" % (nm, cores, tm, n1, t1)}
    json_out.write(json.dumps(res) + "\n")
    json_out.flush()


def ingest_bench(torch, np, orbfe, synth, dev, mode, scene, frames_per_slot, passes=36, slots=3, rgb=False, pageable=False):
    """SURVEY.md 8f-1, the staging half (include/orbfe_ingest.h; buildStream.cpp:376-381, :399-406, :462-466, :483-487):
    the SAME step as the headline (orbfe_extract + orbfe_match_batch), but the frames start in pinned HOST memory and the
    records, counts and matcher outputs end there.  Three figures per slot of `frames_per_slot` frames:
      copy     the slot's upload and download alone, both directions at once (full duplex), no kernels;
      compute  extraction + matching on the slot's device-resident frames, no copies;
      pipelined  the ring running: upload of slot s + 1 / kernels of slot s / download of slot s - 1 overlapped.
    overlap_efficiency = max(copy, compute) / pipelined (1 = perfect overlap; the target is >= 0.9), and the side that
    bounds the configuration is named.  The pinned-copy peaks are measured here, on this box, with plain hipMemcpyAsync
    of 256 MiB buffers (torch pinned tensors), one direction at a time and both at once."""
    m = MODES[mode]
    w, h = m["width"], m["height"]
    mm = m["match"]
    F = frames_per_slot
    ch = 3 if rgb else 1
    ctx = orbfe.Context(w, h, max_batch=F, device=dev.index or 0, **m["cfg"])
    ing = orbfe.Ingest(ctx, F, slots=slots, channels=ch, match_mode=mm["mode"] if mm else -1, match_window=mm["window"] if mm else -1,
                       match_max_distance=mm["max_distance"] if mm else 256, download_matches=2 if mm else 0)
    n_distinct = min(32, F)
    for sl in range(slots):  # every slot holds other scenes
        base = make_scenes(synth, mode, scene, n_distinct, 50000 + 1000 * sl)
        fr = base[np.arange(F) % len(base)]
        if rgb:
            fr = (fr.astype(np.int16)[..., None] + np.array([3, 0, -3], np.int16)).clip(0, 255).astype(np.uint8)
        ing.host_frames(sl)[:] = fr
    pageable_src = np.ascontiguousarray(ing.host_frames(0)).copy() if pageable else None

    def submit(sl):
        if pageable:
            ing.submit_from(sl, pageable_src)
        else:
            ing.submit(sl, F)

    # warm-up: three rounds through the ring (first touch of the pinned pages, the runtime's signal / staging pools, clocks:
    # tools/ingest_probe.py shows the first configuration of a process spending 0.3 ms per submit on the host, 0.06 later)
    for _ in range(3):
        for sl in range(slots):
            submit(sl)
        for sl in range(slots):
            ing.wait(sl)
    t_first = ing.timing(0)
    up_bytes, down_bytes = t_first["upload_bytes"], t_first["download_bytes"]
    # pipelined: `passes` slot passes through the ring, the host only submits and waits
    t0 = time.perf_counter()
    kp = 0
    for i in range(passes):
        sl = i % slots
        if i >= slots:
            _, cnt, _, _ = ing.wait(sl)
            kp += int(cnt.sum())
        submit(sl)
    for i in range(passes, passes + slots):
        sl = i % slots
        _, cnt, _, _ = ing.wait(sl)
        kp += int(cnt.sum())
    t_pipe = (time.perf_counter() - t0) / passes
    per_slot = [ing.timing(sl) for sl in range(slots)]
    up_ms = sum(t["upload_ms"] for t in per_slot) / slots      # inside the running pipeline, from the ring's own events
    cmp_ms = sum(t["compute_ms"] for t in per_slot) / slots
    down_ms = sum(t["download_ms"] for t in per_slot) / slots
    # compute alone: the same calls on the slot's resident device frames, torch's stream, nothing else running
    d_frames, d_rec, d_cnt, d_idx, d_dst = ing.device_buffers(0)
    s = torch.cuda.current_stream().cuda_stream

    def compute_once():
        if rgb:
            ctx.extract_rgb(d_frames, 3 * w, 3 * w * h, F, d_rec, d_cnt, None, s)
        else:
            ctx.extract(d_frames, w, w * h, F, d_rec, d_cnt, None, s)
        if mm:
            ctx.match_batch(d_rec, d_cnt, F, mm["mode"], mm["window"], mm["max_distance"], d_idx, d_dst, s)
    for _ in range(3):
        compute_once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(5, min(200, int(50.0 / max(cmp_ms, 1e-3))))
    e0.record()
    for _ in range(reps):
        compute_once()
    e1.record()
    torch.cuda.synchronize()
    t_compute = e0.elapsed_time(e1) / reps * 1e-3
    ing.close()
    ctx.close()

    # copies alone (pinned torch tensors = hipHostMalloc): peaks at 256 MiB, then this slot's own sizes in both directions at once
    def copy_time(h2d_bytes, d2h_bytes, reps):
        hs = torch.empty(max(h2d_bytes, 1), dtype=torch.uint8).pin_memory()
        ds = torch.empty(max(h2d_bytes, 1), dtype=torch.uint8, device=dev)
        hd = torch.empty(max(d2h_bytes, 1), dtype=torch.uint8).pin_memory()
        dd = torch.zeros(max(d2h_bytes, 1), dtype=torch.uint8, device=dev)
        hs.zero_(); hd.zero_()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        best = None
        for attempt in range(2):  # the first run touches the pages
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                if h2d_bytes:
                    with torch.cuda.stream(s1):
                        ds.copy_(hs, non_blocking=True)
                if d2h_bytes:
                    with torch.cuda.stream(s2):
                        hd.copy_(dd, non_blocking=True)
            torch.cuda.synchronize()
            best = (time.perf_counter() - t0) / reps
        return best
    big = 256 << 20
    peak_h2d = big / copy_time(big, 0, 8) / 1e9
    peak_d2h = big / copy_time(0, big, 8) / 1e9
    t_bidir = copy_time(big, big, 8)
    t_copy = copy_time(up_bytes, down_bytes, max(4, min(64, int(0.2 / max(t_pipe, 1e-4)))))
    bound = max(t_copy, t_compute)
    return {"frames_per_slot": F, "slots": slots, "passes_timed": passes, "input": "RGB8" if rgb else "gray",
            "source": ("pageable host memory, copied into the pinned slot by the submitting thread (orbfe_ingest_submit_from, what the "
                       "driver does behind the reference's cudaMemcpy2DAsync)" if pageable else
                       "pinned slots filled in place by the producer (orbfe_ingest_host_frames)"),
            "value": kp / (t_pipe * passes), "unit": "keypoints/s", "frames_per_s": F / t_pipe,
            "ms_per_slot": {"pipelined": t_pipe * 1e3, "copy_alone_both_directions": t_copy * 1e3, "compute_alone": t_compute * 1e3,
                            "in_pipeline_upload": up_ms, "in_pipeline_compute": cmp_ms, "in_pipeline_download": down_ms},
            "bytes_per_slot": {"upload": up_bytes, "download": down_bytes},
            "h2d_GBps": {"achieved_pipelined": up_bytes / t_pipe / 1e9, "achieved_while_copying": up_bytes / (up_ms * 1e-3) / 1e9,
                         "peak_measured_pinned": peak_h2d, "frac_of_peak": up_bytes / t_pipe / 1e9 / peak_h2d},
            "d2h_GBps": {"achieved_pipelined": down_bytes / t_pipe / 1e9, "peak_measured_pinned": peak_d2h},
            "bidirectional_GBps_measured_pinned": 2 * big / t_bidir / 1e9,
            # (the timed region includes the ring's fill and drain: one upload + compute + download of a slot that overlaps nothing,
            # i.e. about (compute + download) / passes per slot on top of the steady state)
            "overlap_efficiency": bound / t_pipe, "bound_by": "pcie_h2d" if t_copy >= t_compute else "compute",
            "copy_over_compute": t_copy / t_compute,
            "note": "same step as `value` of the headline but host -> device -> host; overlap_efficiency = max(copy alone, compute "
                    "alone) / pipelined per slot; peaks are plain pinned hipMemcpyAsync of 256 MiB measured in this run"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=sorted(MODES) + ["match", "align"], default="c2")
    ap.add_argument("--batch", type=int, default=0, help="frames per GPU per step (c2/ref/c3), total frames (c4); "
                                                           "0 = the mode's default")
    ap.add_argument("--scene", choices=("dense", "survey"), default="dense")
    ap.add_argument("--distinct", type=int, default=64, help="distinct synthetic frames per rank (repeated to fill the batch)")
    ap.add_argument("--prewarm", type=int, default=-1,
                    help="untimed steps issued before the warm-up steps so that the clocks have settled (not counted); "
                         "-1 = about 50 ms worth of them")
    ap.add_argument("--rotate", type=int, default=1,
                    help="cycle this many distinct resident input batches (step i reads batch i %% k): shows the rate "
                         "does not depend on a step finding the previous step's data in cache")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (survey scene, no-gather rate)")
    ap.add_argument("--rgb", action="store_true",
                    help="feed interleaved RGB8 frames (SURVEY.md 8f-1): the gray conversion is fused into "
                         "the pyramid kernel; not the BASELINE metric (its configs are grayscale)")
    ap.add_argument("--ingest", action="store_true",
                    help="SURVEY.md 8f-1 staging: print ONE JSON line for the host -> device -> host pipeline (include/orbfe_ingest.h) "
                         "at several slot sizes, gray and RGB8, pinned and pageable sources; 1 GPU.  The default line carries the "
                         "1024-frame gray figure as the key `ingest`")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: leave the keypoint gather out of the timed region")
    ap.add_argument("--exact-gather", action="store_true", help="N > 1: variable-length gather (counts first, then exactly "
                                                                  "sum(counts) * 52 bytes per rank); = --gather exact")
    ap.add_argument("--gather", choices=("auto", "fixed", "exact"), default="auto",
                    help="N > 1: which form of the keypoint gather.  fixed: cap records per frame whatever the counts, nothing touches "
                         "the host; exact: counts first, then only the valid records (one host read of the counts per step); auto "
                         "(default): exact when the frames fill less than three quarters of their record capacity (decided from one "
                         "untimed extraction, the same on every rank), fixed otherwise")
    ap.add_argument("--gather-every", type=int, default=1,
                    help="N > 1: ship the records of every k-th step only (a consumer that samples the stream: the link carries 1 / k "
                         "of the bytes).  1 = every step, what the BASELINE metric means; reported in the line when not 1")
    ap.add_argument("--dry-run", action="store_true", help="with --gpus N: print the child command lines and exit")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gather_every < 1:
        ap.error("--gather-every must be >= 1")
    if args.exact_gather:
        args.gather = "exact"

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # this process has not imported torch or touched HIP: start the ranks as children
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    if args.dry_run:
        print(json.dumps([]))
        return
    # stdout carries exactly ONE line, the JSON: whatever libraries print there (gloo's connection
    # banner, RCCL notices) is sent to stderr for the rest of the run
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    import orbfe
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE is %d (launch with --nproc-per-node %d or drop WORLD_SIZE)"
                 % (args.gpus, world, args.gpus))
    if not os.path.exists(orbfe.LIB_PATH) and local_rank == 0:
        import __graft_entry__  # clean checkout: compile the HIP library first (hipcc, gfx950)
        __graft_entry__.build()
    from orbfe import synth
    from orbfe.dist import RcclComm, gather_keypoints_async, merge_cell_keys, shard_range

    share = os.environ.get("ORBFE_BENCH_SHARE_GPU") == "1"  # rehearsal on a 1-GPU box: every rank on cuda:0, gloo
    # ORBFE_BENCH_FORCE_COMM=1: run the N > 1 code path (torch.distributed nccl + the C++/RCCL communicator
    # and the gather inside the timed region) with a single rank -- everything but the xGMI transport
    force_comm = os.environ.get("ORBFE_BENCH_FORCE_COMM") == "1" and world == 1
    multi = world > 1 or force_comm
    comm = None
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share:
            local_rank = 0
        elif torch.cuda.device_count() < world:
            sys.exit("bench.py: %d ranks but %d HIP devices (ORBFE_BENCH_SHARE_GPU=1 rehearses on one)"
                     % (world, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if multi and not share:
        def exchange(ident):  # rank 0's RCCL unique id reaches the other ranks through torch.distributed
            t = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                t.copy_(torch.frombuffer(bytearray(ident), dtype=torch.uint8))
            dist.broadcast(t, 0)
            return bytes(t.cpu().numpy().tobytes())
        try:
            comm = RcclComm(rank, world, local_rank, exchange)
        except Exception as e:  # keep the run alive on torch.distributed's own RCCL gather, and say so
            sys.stderr.write("bench.py: liborbfe_dist.so communicator unavailable (%s); using torch.distributed\n" % e)
            comm = None
        ok = torch.tensor([1 if comm is not None else 0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # all ranks take the same path
        if int(ok.item()) == 0 and comm is not None:
            comm.close()
            comm = None

    if args.mode == "match":
        if world != 1:
            sys.exit("bench.py: --mode match is a single-GPU microbench")
        match_microbench(args, torch, np, orbfe, dev, json_out)
        return
    if args.mode == "align":
        if world != 1:
            sys.exit("bench.py: --mode align is a single-GPU bench (frames are independent: N GPUs = N replicas)")
        align_bench(args, torch, np, orbfe, dev, json_out)
        return
    if args.ingest:
        if world != 1:
            sys.exit("bench.py: --ingest is a single-GPU bench (every GPU has its own PCIe link: N GPUs = N replicas)")
        mode = args.mode if args.mode in ("c2", "ref") else "c2"  # the ring matches f - 1 -> f inside a slot, as these modes do
        runs = []
        for F, rgb, pageable in ((256, False, False), (1024, False, False), (4096, False, False), (1024, True, False),
                                 (1024, False, True)):
            F = min(F, args.batch) if args.batch else F
            runs.append(ingest_bench(torch, np, orbfe, synth, dev, mode, args.scene, F, passes=max(args.steps, 36) if F < 4096 else 18,
                                     rgb=rgb, pageable=pageable))
        head = next(r for r in runs if r["frames_per_slot"] == (min(1024, args.batch) if args.batch else 1024) and r["input"] == "gray"
                    and r["source"].startswith("pinned"))
        mm_ = MODES[mode]
        line = {"metric": "ORB keypoints/sec end-to-end INCLUDING PCIe (pinned host frames in, host records out), %dx%d" % (mm_["width"], mm_["height"]),
                "value": head["value"], "unit": "keypoints/s", "n_gpus": 1, "steps": head["passes_timed"], "warmup": 3,
                "ms_per_step": head["ms_per_slot"]["pipelined"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "u8", "data": "synthetic",
                "config": {"workload": mm_["workload"] + " -- staged through include/orbfe_ingest.h", "mode": mode, "ingest": True,
                           "frames_per_gpu_per_step": head["frames_per_slot"]},
                "roofline": {"bound": "pcie", "achieved": head["h2d_GBps"]["achieved_pipelined"], "peak": head["h2d_GBps"]["peak_measured_pinned"],
                             "unit": "GB/s", "frac": head["h2d_GBps"]["frac_of_peak"], "traffic": None,
                             "note": "this line is bound by the host link, not by HBM: achieved = uploaded bytes / pipelined time, peak = "
                                     "pinned hipMemcpyAsync H2D measured in the same run; the device-resident HBM roofline is the default line's"},
                "overlap_efficiency": head["overlap_efficiency"], "bound_by": head["bound_by"], "runs": runs, "cpu_baseline": None}
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()
        return
    m = MODES[args.mode]
    w, h = m["width"], m["height"]
    total = args.batch or m["batch"]
    if m["scaling"] == "strong":
        f0, f1 = shard_range(total, rank, world) if args.mode != "c5" else (0, 1)
        B = f1 - f0
    else:
        f0, B = 0, total
    if args.mode == "c3" and B % 2:
        sys.exit("bench.py: c3 needs an even number of frames (stereo pairs)")
    mm = m["match"]
    s = torch.cuda.current_stream().cuda_stream

    def build_input(scene, variant=0):
        n_distinct = max(min(args.distinct, B), 1)
        if args.mode == "c3":
            n_distinct += n_distinct % 2
        # every rank sees different scenes; strong-scaling modes index them by global frame number
        first = (f0 if m["scaling"] == "strong" else 1000 * rank) + 100000 * variant
        base = make_scenes(synth, args.mode, scene, n_distinct, first)
        fr = torch.from_numpy(base).to(dev)[torch.arange(B, device=dev) % len(base)].contiguous()
        if args.rgb:  # R = G = B = gray scene +- a channel-dependent offset: corners survive the conversion
            off = torch.tensor([3, 0, -3], dtype=torch.int16, device=dev)
            fr = (fr.to(torch.int16).unsqueeze(-1) + off).clamp(0, 255).to(torch.uint8).contiguous()
        return base, fr

    ctx = orbfe.Context(w, h, max_batch=max(B, 1), device=local_rank, **m["cfg"]) if B > 0 else None
    cap = ctx.cap if ctx else 0
    base, frames = build_input(args.scene)
    # --gather auto: one untimed extraction tells how full the records are; every rank must take the same form, so the
    # decision is made on the global mean
    gather_fill = None
    if multi and args.mode != "c5" and args.gather == "auto":
        kp_frames = torch.zeros(2, dtype=torch.float64, device=dev)
        if B > 0:
            r0 = torch.zeros(B * cap * 52, dtype=torch.uint8, device=dev)
            c0 = torch.zeros(B, dtype=torch.int32, device=dev)
            if args.rgb:
                ctx.extract_rgb(frames.data_ptr(), 3 * w, 3 * w * h, B, r0.data_ptr(), c0.data_ptr(), None, s)
            else:
                ctx.extract(frames.data_ptr(), w, w * h, B, r0.data_ptr(), c0.data_ptr(), None, s)
            torch.cuda.synchronize()
            kp_frames[0], kp_frames[1] = float(c0.sum().item()), float(B)
            del r0, c0
        t = kp_frames.cpu() if share else kp_frames
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        gather_fill = float(t[0].item()) / max(float(t[1].item()) * max(cap, 1), 1.0)
        args.exact_gather = gather_fill < 0.75
    elif args.gather == "fixed":
        args.exact_gather = False
    n_pairs = max((B - 2) // m["stride"] + 1, 0) if (mm and B >= 2) else 0
    # records / counts are double-buffered: the gather of step i overlaps the kernels of step i + 1
    recs = [torch.zeros(max(B, 1) * max(cap, 1) * 52, dtype=torch.uint8, device=dev) for _ in range(2)]
    cnts = [torch.zeros(max(B, 1), dtype=torch.int32, device=dev) for _ in range(2)]
    idx = torch.zeros(max(n_pairs, 1) * max(cap, 1), dtype=torch.int32, device=dev)
    dst = torch.zeros(max(n_pairs, 1) * max(cap, 1), dtype=torch.int32, device=dev)
    keys = torch.zeros(max(ctx.K if ctx else 1, 1), dtype=torch.int32, device=dev) if args.mode == "c5" else None
    equal_shards = m["scaling"] == "weak" or total % world == 0
    gather_ok = multi and args.mode != "c5" and equal_shards
    gather_out = None
    if gather_ok and rank == 0:
        gather_out = [(torch.empty((world, B * cap * 52), dtype=torch.uint8, device=dev),
                       torch.empty((world, B), dtype=torch.int32, device=dev)) for _ in range(2)]
        if not args.exact_gather:  # rank 0 extracts straight into its own block of the gathered arrays: no local copy
            recs = [gather_out[b][0][0] for b in range(2)]
            cnts = [gather_out[b][1][0] for b in range(2)]
    state = dict(frames=None, step=0, tickets=[0, 0], pending=[None, None], events=[], prewarm=0, ctx=ctx)

    def run_step(record, do_gather):
        ctx = state["ctx"]  # the context of this run (extras time other configurations on the same buffers)
        b = state["step"] & 1
        frames = state["frames"][state["step"] % len(state["frames"])]  # --rotate: another resident batch every step
        state["step"] += 1
        do_gather = do_gather and (state["step"] - 1) % args.gather_every == 0  # --gather-every k: steps 0, k, 2k, ...
        if do_gather:  # the gather that last read this buffer pair must be done before it is rewritten
            if comm is not None:
                comm.wait_ticket(state["tickets"][b], s)
            elif state["pending"][b] is not None:
                state["pending"][b].wait()
                state["pending"][b] = None
        r, c = recs[b], cnts[b]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if record else None
        # orbfe_extract == build_pyramid + detect_batch + describe_batch; issued separately so
        # that HIP events on this stream can time each stage inside the timed region
        if ev: ev[0].record()
        if B > 0:
            if args.rgb:
                orbfe.check(orbfe.lib().orbfe_build_pyramid_rgb(ctx.handle, frames.data_ptr(), 3 * w, 3 * w * h, B, s),
                            ctx.handle)
            else:
                ctx.build_pyramid(frames.data_ptr(), w, w * h, B, s)
        if ev: ev[1].record()
        if args.mode == "c5":
            ctx.detect_batch_shard(B, rank, world, s)
            if multi:  # partial per-cell keys -> element-wise MAX over the ranks -> back into the context
                ctx.export_cell_keys(B, keys.data_ptr(), s)
                if comm is not None:
                    comm.allreduce_max_keys(keys.data_ptr(), ctx.K, s)
                    comm.wait(s)
                else:
                    merge_cell_keys(keys)
                ctx.import_cell_keys(B, keys.data_ptr(), s)
        elif B > 0:
            ctx.detect_batch(B, s)
        if ev: ev[2].record()
        if B > 0:
            ctx.describe_batch(B, r.data_ptr(), c.data_ptr(), None, s)
        if ev: ev[3].record()
        if n_pairs > 0:
            ctx.match_pairs(r.data_ptr(), c.data_ptr(), B, 0, m["stride"], mm["mode"], mm["window"], mm["max_distance"],
                            idx.data_ptr(), dst.data_ptr(), s)
        if ev:
            ev[4].record()
            state["events"].append(ev)
        if do_gather:
            if comm is not None:
                go = gather_out[b] if gather_out else (None, None)
                comm.gather_keypoints(r.data_ptr(), c.data_ptr(), B, cap, go[0].data_ptr() if go[0] is not None else None,
                                      go[1].data_ptr() if go[1] is not None else None, 0, args.exact_gather, s)
                state["tickets"][b] = comm.ticket()
            else:
                state["pending"][b] = gather_keypoints_async(r, c, gather_out[b] if gather_out else None, dst=0)

    def sync():
        for b in (0, 1):
            if state["pending"][b] is not None:
                state["pending"][b].wait()
                state["pending"][b] = None
        if comm is not None:
            comm.sync()  # "gather complete on rank 0"
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(frames, do_gather, record, steps=None):
        steps = steps or args.steps
        state["frames"] = frames if isinstance(frames, list) else [frames]
        state["events"] = []
        # the clocks take ~50 ms of load to settle (measured: 0.522 ms per step after 3 warm-up steps, 0.497
        # after 20, 0.484 after 100, 0.482 over 200 timed steps): settle first, then the W warm-up steps
        for _ in range(state["prewarm"]):
            run_step(False, do_gather)
        sync()
        for _ in range(args.warmup):
            run_step(False, do_gather)
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            # per-stage HIP events on every 4th timed step: five event records per step cost ~3 % of the
            # step (0.482 ms with them on every step, 0.466 without any: tools/match_overlap_probe.py)
            run_step(record and i % 4 == 0, do_gather)
        sync()
        elapsed = time.perf_counter() - t0
        counts = cnts[(state["step"] - 1) & 1].cpu().numpy().astype(np.int64)[:B]
        kp_local = int(counts.sum())
        prs = [int(counts[k * m["stride"]] * counts[k * m["stride"] + 1]) for k in range(n_pairs)]
        pairs_local = int(sum(prs))
        if multi:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev if not share else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            tot = torch.tensor([kp_local, pairs_local, B], dtype=torch.int64, device=dev if not share else "cpu")
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            kp_total, pairs_total, frames_total = (int(v) for v in tot.tolist())
        else:
            kp_total, pairs_total, frames_total = kp_local, pairs_local, B
        if args.mode == "c5":  # every rank describes the same merged frame: count it once
            kp_total, frames_total = kp_local, 1
        return dict(elapsed=elapsed, counts=counts, kp_local=kp_local, pairs_local=pairs_local, kp_total=kp_total,
                    pairs_total=pairs_total, frames_total=frames_total, events=state["events"], steps=steps)

    use_gather = gather_ok and not args.no_gather
    if args.rotate > 1:
        frames = [frames] + [build_input(args.scene, v)[1] for v in range(1, args.rotate)]
    # clock settling: ~50 ms of load before the warm-up steps (0.522 ms per 256-frame step after 3 warm-up steps,
    # 0.484 after 100); a step's length is not known yet, so estimate it from the frame count
    state["prewarm"] = args.prewarm if args.prewarm >= 0 else max(4, min(100, int(100 * 256 * 307200 / max(B * w * h, 1))))
    main_run = timed(frames, use_gather, True)
    extras = {}
    if not args.no_extras:
        # the same step sustained for about three seconds (no stage events): the K-step figure above is not a
        # burst, and whoever samples the GPU from outside every few seconds sees it busy
        n_long = max(args.steps, min(60000, int(3.0 / max(main_run["elapsed"] / args.steps, 1e-6))))
        r1 = timed(frames, use_gather, False, steps=n_long)
        extras["sustained"] = {"steps": n_long, "value": r1["kp_total"] * n_long / r1["elapsed"], "unit": "keypoints/s",
                               "ms_per_step": r1["elapsed"] / n_long * 1e3, "seconds": r1["elapsed"]}
        if use_gather:  # the same steps without the collective
            r2 = timed(frames, False, False)
            extras["no_gather"] = {"value": r2["kp_total"] * args.steps / r2["elapsed"], "unit": "keypoints/s",
                                   "ms_per_step": r2["elapsed"] / args.steps * 1e3}
        other = "survey" if args.scene == "dense" else "dense"
        _, frames2 = build_input(other)
        r3 = timed(frames2, use_gather, False)
        extras[other + "_scene"] = {"value": r3["kp_total"] * args.steps / r3["elapsed"], "unit": "keypoints/s",
                                    "ms_per_step": r3["elapsed"] / args.steps * 1e3,
                                    "frames_per_s": r3["frames_total"] * args.steps / r3["elapsed"],
                                    "keypoints_per_frame": r3["kp_total"] / max(r3["frames_total"], 1)}
        del frames2
        if world == 1 and args.mode in ("c2", "ref") and not args.rgb:
            # f1's staging half: the same step with the frames starting in pinned host memory and the results ending there
            try:
                extras["ingest"] = ingest_bench(torch, np, orbfe, synth, dev, args.mode, args.scene, min(1024, max(B, 2)), passes=36)
            except Exception as e:  # never lose the headline over the extra
                extras["ingest"] = {"error": str(e)}
        if args.mode == "c2" and world == 1:
            # The corrected EXT modes on the same step.  The headline is the reference's quirk-parity
            # regime: orientation in radians used as degrees (Q7), description always on level 0 (Q10).  The 222-break
            # orientation table of the tile describe kernel exists ONLY there (|theta| <= 0.055 rad); angle_in_radians = 1
            # (real rotation) runs the arithmetic loop with a 19-px halo, descriptor_level = 1 (scale-aware) describes on the
            # level that won the cell.  Same frames, same buffers, one context at a time.
            fixed = {}
            for name, kw in (("descriptor_level", dict(descriptor_level=1)), ("angle_in_radians", dict(angle_in_radians=1)),
                             ("both", dict(descriptor_level=1, angle_in_radians=1))):
                state["ctx"] = None
                c2 = orbfe.Context(w, h, max_batch=max(B, 1), device=local_rank, **dict(m["cfg"], **kw))
                state["ctx"] = c2
                rr = timed(frames, False, True)
                st_ms = {k: 0.0 for k in ("pyramid", "detect", "describe", "match")}
                for ev in rr["events"]:
                    for i, k in enumerate(st_ms):
                        st_ms[k] += ev[i].elapsed_time(ev[i + 1])
                fixed[name] = {"value": rr["kp_total"] * args.steps / rr["elapsed"], "unit": "keypoints/s",
                               "ms_per_step": rr["elapsed"] / args.steps * 1e3,
                               "stage_ms": {k: v / max(len(rr["events"]), 1) for k, v in st_ms.items()},
                               "keypoints_per_frame": rr["kp_total"] / max(rr["frames_total"], 1),
                               "kernels": c2.dispatch_info(max(B, 1), mm["mode"], mm["window"])["describe"]}
                c2.close()
            state["ctx"] = ctx
            fixed["note"] = ("the headline `value` is the reference's quirk-parity regime (angle_in_radians = 0, descriptor_level "
                             "= 0); these are the same step with the quirks fixed (orbfe_config), timed the same way")
            extras["fixed_modes"] = fixed

    out = None
    if rank == 0:
        R = main_run
        names = ("pyramid", "detect", "describe", "match")
        ms = {k: 0.0 for k in names}
        for ev in R["events"]:
            for i, k in enumerate(names):
                ms[k] += ev[i].elapsed_time(ev[i + 1])
        stages = {k: ms[k] / max(len(R["events"]), 1) for k in names}
        cfg = m["cfg"]
        detect_levels = sum(1 for l in range(cfg["levels"]) if (cfg["cell"] >> l) > 0 and (w >> l) > 0 and (h >> l) > 0)
        k_out = R["kp_local"] / max(B, 1)
        ab = algorithmic_bytes(w, h, cfg["levels"], detect_levels, ctx.K, k_out)
        c64 = R["counts"]
        st = m["stride"]
        if n_pairs:
            a_, b_ = c64[0:n_pairs * st:st], c64[1:n_pairs * st + 1:st]
            per = (40 if mm["mode"] == 1 else 12)
            ab["match"] = float((per * (a_ + b_) + 8 * a_).sum()) / B  # descriptor + position in, (idx, dist) out
        else:
            ab["match"] = 0.0
        # the kernels this context actually runs for this call size (the library picks the describe and match kernels)
        disp = ctx.dispatch_info(max(B, 1), mm["mode"] if mm else 1, mm["window"] if mm else -1)
        if not mm:
            disp["match"], disp["match_examines"] = None, None
        mfma_match = bool(mm) and ("match_mfma_kernel" in (disp["match"] or "") or "match_tile_kernel" in (disp["match"] or ""))
        kernels = {k: disp[k] for k in names}
        # PMC-derived numbers (profiles/traffic.json: rocprofv3 --pmc passes, tools/collect_profiles.sh) are used only
        # when they were measured on THIS source (orbfe.source_hash()); they are per launch of `batch` frames and scale
        # linearly with the frames of a launch
        prof, pmc = {}, {"source": "profiles/traffic.json", "csrc_sha256_built": orbfe.source_hash()}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        try:
            prof = json.load(open(tpath)).get(args.mode, {})
        except Exception:
            prof = {}
        pmc["csrc_sha256_measured"] = prof.get("csrc_sha256")
        pmc["stale"] = not prof or prof.get("csrc_sha256") != pmc["csrc_sha256_built"]
        if pmc["stale"]:
            pmc["note"] = ("no PMC passes for these sources: traffic / VALU fields are null (re-run tools/collect_profiles.sh "
                           "and commit profiles/traffic.json)")
            prof = {}
        scale = B / float(prof.get("batch", 256)) if prof else 0.0
        valu_counts = prof.get("valu_wave_instructions", {})
        per_stage = {}
        for k in names:
            t = stages[k] * 1e-3
            if t <= 0:
                continue
            e = {"kernel": kernels[k], "ms": stages[k], "algorithmic_bytes": ab[k] * B,
                 "achieved_GBps": ab[k] * B / t / 1e9, "frac": ab[k] * B / t / 1e9 / HBM_PEAK_GBPS,
                 "traffic": prof[k] * scale if prof.get(k) else None}
            if valu_counts.get(k):  # VALU wave-instructions per launch (rocprofv3 SQ_INSTS_VALU) x 64 lanes
                ach = valu_counts[k] * scale * 64 / t / 1e12
                e["valu"] = {"bound": "valu", "unit": "Tlane-op/s", "achieved": ach, "peak": VALU_PEAK_TLANEOPS,
                             "frac": ach / VALU_PEAK_TLANEOPS, "wave_instructions_per_launch": valu_counts[k] * scale,
                             "lane_ops_per_level0_pixel": valu_counts[k] * scale * 64 / (B * w * h)}
                # SQ_ACTIVE_INST_VALU x 4 / SIMD-cycles: the counter charges every vector instruction one quad-cycle, but a
                # full-rate instruction retires in 2.76 cycles on this chip, so this is an ISSUE COUNT per
                # SIMD-cycle, not a busy fraction: it can exceed 1 (detect: 1.06)
                busy = prof.get("valu_issue_quad_cycles_per_simd_cycle", prof.get("valu_pipe_busy", {})).get(k)
                if busy is not None:
                    e["valu"]["issue_quad_cycles_per_simd_cycle"] = busy
            per_stage[k] = e
        if mfma_match and "match" in per_stage:
            # the 256-bit matcher runs on the matrix cores: 2 x 256 flop per pair on e2m1 operands,
            # dense FP4 MFMA peak ~10 PFLOP/s (MI355X_MICROARCH.md, Matrix cores)
            flops = 512.0 * R["pairs_local"]
            t = stages["match"] * 1e-3
            per_stage["match"]["mfma"] = {"bound": "mfma", "unit": "TFLOP/s", "peak": FP4_MFMA_PEAK_TFLOPS,
                                          "achieved": flops / t / 1e12, "frac": flops / t / 1e12 / FP4_MFMA_PEAK_TFLOPS,
                                          "flops_per_launch": flops}
        dom = max(per_stage, key=lambda k: stages[k])  # the kernel with the largest share of the step
        # The tier's declared roofline is HBM: `roofline.frac` = algorithmic bytes of the dominant kernel / its launch
        # time / 8 TB/s.  What the counters show limits that kernel (vector-instruction issue for detect and describe,
        # the matrix cores for the brute-force matcher) sits beside it as named sub-objects, never in `frac`.
        # `limiter` is read from the counters of THIS build, not assumed from the stage's name: the largest of
        #   hbm  = PMC traffic / launch time / 6.29 TB/s (what a plain copy reaches on this chip),
        #   valu = vector-instruction issue quad-cycles per SIMD-cycle / its ceiling 1.45,
        #   mfma = matrix-core flop fraction of the dense FP4 peak;  null when there are no PMC passes for this build
        # VALU issue: the counter charges every vector instruction one quad-cycle (4 cycles) whatever it costs; a stream of
        # full-rate instructions retires one per 2.76 cycles on this chip (tools/valu_rate5.hip), so the counter's CEILING is
        # 4 / 2.76 = 1.45 per SIMD-cycle (0.96 for a stream of 4.15-cycle instructions): the utilisation is the raw value over
        # that ceiling, both are printed, and a limiter is named only when it leads the runner-up by more than the
        # calibration error (0.1) -- otherwise "a~b" 
        VALU_ISSUE_CEILING = 4.0 / 2.76

        def limiter_of(e):
            util = {}
            if e.get("traffic"):
                util["hbm"] = e["traffic"] / (e["ms"] * 1e-3) / 1e9 / 6290.0
            v = (e.get("valu") or {}).get("issue_quad_cycles_per_simd_cycle")
            if v is not None:
                util["valu"] = v / VALU_ISSUE_CEILING
                e["valu"]["issue_ceiling"] = VALU_ISSUE_CEILING
                e["valu"]["issue_utilisation"] = v / VALU_ISSUE_CEILING
            if e.get("mfma"):
                util["mfma"] = e["mfma"]["frac"]
            if "hbm" not in util or "valu" not in util:
                return None, util
            order = sorted(util, key=util.get, reverse=True)
            if len(order) > 1 and util[order[0]] - util[order[1]] <= 0.1:
                return order[0] + "~" + order[1], util
            return order[0], util
        for e in per_stage.values():
            e["limiter"], e["limiter_utilisation"] = limiter_of(e)
        limiter = per_stage[dom]["limiter"]
        roof = {"bound": "hbm", "achieved": per_stage[dom]["achieved_GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": per_stage[dom]["frac"], "traffic": per_stage[dom]["traffic"],
                "kernel": kernels[dom], "stage": dom, "algorithmic_bytes_per_launch": ab[dom] * B,
                "avg_launch_ms": stages[dom], "frames_per_launch": B,
                "limiter": limiter, "valu": per_stage[dom].get("valu"), "mfma": per_stage[dom].get("mfma"),
                "pmc": pmc,
                "note": "frac = algorithmic bytes / launch time / 8 TB/s for the kernel with the largest share of the step "
                        "(HIP events on the work stream inside the timed region); `limiter` = the most utilised of HBM (PMC traffic "
                        "vs the 6.29 TB/s a copy reaches), VALU issue and MFMA, from the PMC passes of this build (null without "
                        "them), with that roof in `valu` / `mfma` (DESIGN.md section 4)",
                "stages": per_stage}
        elapsed = R["elapsed"]
        ms_step = elapsed / args.steps * 1e3
        out = {
            "metric": "ORB keypoints/sec end-to-end (extract + match), %dx%d %d-level" % (w, h, cfg["levels"]),
            "value": R["kp_total"] * args.steps / elapsed,
            "unit": "keypoints/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "timed_region_ms": elapsed * 1e3,
            "higher_is_better": True, "scaling": m["scaling"], "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": m["workload"] + (" [RGB8 input, conversion fused]" if args.rgb else ""),
                       "mode": args.mode,
                       "scene": ("dense: 800 rectangles of 6..32 px per 640x480 of area + noise (fills the feature budget)"
                                 if args.scene == "dense" else
                                 "survey: SURVEY.md 8d generator, 96 rectangles up to a fifth of the frame + -3..3 noise"),
                       "distinct_frames_per_rank": int(len(base)) * max(args.rotate, 1), "rotate": args.rotate,
                       "prewarm_steps": state["prewarm"],
                       "frames_per_gpu_per_step": B, "frames_per_step": R["frames_total"],
                       "resident_input_bytes_per_gpu": int(B) * w * h * (3 if args.rgb else 1) * max(args.rotate, 1),
                       "keypoints_per_frame": R["kp_total"] / max(R["frames_total"], 1),
                       "collective": (("RCCL (liborbfe_dist.so: grouped ncclSend/ncclRecv, %s) gather of 52-byte keypoint "
                                       "records + counts to rank 0 inside the timed region, overlapped with the next step"
                                       % ("exact length" if args.exact_gather else "fixed stride")) if use_gather and comm
                                      else "torch.distributed gather (%s)" % ("gloo, shared-GPU rehearsal" if share else "nccl = RCCL") if use_gather
                                      else ("RCCL all-reduce(MAX) of the per-cell keys (liborbfe_dist.so)" if comm is not None else
                                            "torch.distributed all-reduce(MAX) of the per-cell keys (%s)" % ("gloo, shared-GPU rehearsal" if share else "nccl = RCCL"))
                                      if (args.mode == "c5" and multi)
                                      else "none in the data path; barrier + timing reductions only")},
            "frames_per_s": R["frames_total"] * args.steps / elapsed,
            # descriptor pairs compared per second: only where the matcher really examines every pair (the cell-indexed
            # windowed forms look at ~10 candidates per query; nA * nB would overstate them ~200x)
            "matcher_gpairs_per_s": ((R["pairs_local"] / (stages["match"] * 1e-3) / 1e9)
                                     if stages["match"] > 0 and disp.get("match_examines") == "all_pairs" else None),
            "matcher_examines": disp.get("match_examines"),
            "matcher_pairs_per_step": R["pairs_total"],
            "stage_ms": stages,
            # whole path against the HBM roofline: algorithmic bytes of extraction + matching per step / the WHOLE step
            "path_hbm": {"algorithmic_bytes_per_frame": ab["frame"] + ab["match"],
                         "achieved_GBps": (ab["frame"] + ab["match"]) * B / (ms_step * 1e-3) / 1e9,
                         "frac_of_8TBps": (ab["frame"] + ab["match"]) * B / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "frac_of_6.29TBps_measured_copy": (ab["frame"] + ab["match"]) * B / (ms_step * 1e-3) / 1e9 / 6290.0,
                         "time": "ms_per_step (the whole step, matcher included; rank 0's bytes / the max-over-ranks time)"},
            "roofline": roof,
        }
        if multi:
            # what the gather moves, so that the first real multi-GPU run explains itself: xGMI is point to point, every
            # non-root rank ships its block over its own link to the root (7 links x ~153 GB/s peak per GPU)
            rec_bytes = ((int(R["counts"].sum()) if args.exact_gather else B * cap) * 52 + 4 * B) / float(args.gather_every)
            out["gather"] = {"form": "exact" if args.exact_gather else "fixed_stride",
                             "bytes_shipped_per_nonroot_rank_per_step": rec_bytes if use_gather else 0,
                             "root_ingress_bytes_per_step": rec_bytes * (world - 1) if use_gather else 0,
                             "per_link_GBps_needed_at_this_step_time": (rec_bytes / (ms_step * 1e-3) / 1e9) if use_gather else 0.0,
                             # a gather moves data ONE way over each non-root rank's own link to the root: 153.6 GB/s per
                             # link is the bidirectional figure, a direction offers half of it
                             "xgmi_link_peak_GBps_one_way": 76.8, "xgmi_link_peak_GBps_bidirectional": 153.6, "links_per_gpu": 7,
                             "link_utilisation_one_way": ((rec_bytes / (ms_step * 1e-3) / 1e9) / 76.8) if use_gather else 0.0,
                             "note": "fixed stride ships cap records per frame whatever the counts; the root's own block is written "
                                     "in place (no copy); with --gather-every k the bytes are the per-step average.  Hardware status: the world > 1 "
                                     "branch of liborbfe_dist.so runs in the GPU tests over a loopback stand-in for librccl (2 and 3 ranks on "
                                     "one device, tests/fake_rccl); the xGMI transport itself has not run on a multi-GPU node of the build pool"}
        if not multi and m["scaling"] == "weak" and B > 0:
            # What the link arithmetic predicts for the driver's N = 2, 4, 8 runs of this mode, so that the first run on a
            # multi-GPU node is a check, not a discovery.  Model: the gather of step i rides under the
            # kernels of step i + 1 (double-buffered records), every non-root rank ships its block over ITS OWN xGMI link to
            # the root (76.8 GB/s one way; <= 7 peers, so the root's links never share), hence a steady-state step takes
            # max(compute, bytes per rank / link rate) whatever N is, and weak-scaling efficiency = compute / that.  Two link
            # rates: the 76.8 GB/s peak and 0.8 of it (what a send / recv pair is assumed to sustain; unmeasured here).
            fixed_b = B * cap * 52 + 4 * B
            exact_b = int(R["kp_local"]) * 52 + 4 * B
            fill = R["kp_local"] / max(B * cap, 1)
            auto_form = "exact" if fill < 0.75 else "fixed"

            def predict(nbytes):
                o = {"bytes_per_nonroot_rank_per_step": nbytes}
                for tag, rate in (("at_link_peak", 76.8e9), ("at_0.8_of_link_peak", 0.8 * 76.8e9)):
                    t_link = nbytes / rate * 1e3
                    o[tag] = {"link_ms_per_step": t_link, "step_ms": max(ms_step, t_link),
                              "weak_scaling_efficiency": ms_step / max(ms_step, t_link), "bound": "link" if t_link > ms_step else "compute"}
                return o
            out["scaling_prediction"] = {
                "n_gpus": [2, 4, 8], "compute_ms_per_step": ms_step, "record_fill": fill, "auto_form": auto_form,
                "fixed_stride": predict(fixed_b), "exact_length": predict(exact_b),
                "gather_every_2_fixed_stride": predict(fixed_b / 2.0),
                "note": "same prediction for N = 2, 4 and 8 (point-to-point links: every peer has its own to the root); the exact form "
                        "adds one host read of the counts per step, not modelled; --gather auto picks `auto_form`"}
        if multi:
            out.setdefault("gather", {})["every"] = args.gather_every
            out["gather"]["record_fill_probe"] = gather_fill
        out.update(extras)
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is timed at N = 1 only
            out["cpu_baseline"] = cpu_baseline(base, args.mode)
        else:
            out["cpu_baseline"] = None
    if comm is not None:
        comm.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()


if __name__ == "__main__":
    main()
