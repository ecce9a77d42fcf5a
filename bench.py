#!/usr/bin/env python3
"""bench.py -- ORB front-end throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--mode c2|ref]

One "step" = one pass of the hot path over one batch of synthetic frames already resident
in HBM: orbfe_extract (blur + pyramid -> fused FAST/NMS -> selection -> orientation + rBRIEF
-> 52-byte records) followed by orbfe_match_batch (frame t-1 -> t inside the batch).  Frames are
independent, so for N > 1 they shard across ranks (weak scaling: --batch frames per GPU) and
the data path has no collective; RCCL carries only the barrier and the final reductions of the
timing.  --gather adds the optional collection of every rank's records on rank 0 (RCCL gather,
asynchronous, overlapped with the next step).  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): 640x480 mono, 8-level pyramid, 2000 features/frame.
In this code base that is: cell 8 (4800 cells, detection on levels 0..3 where the cell is
>= 1 px), FAST-9 with t = 13, top-2000 by (score desc, cell asc), 256-bit brute-force
matching.  --mode ref runs the reference-parity configuration instead (cell 32, FAST-12,
6 levels, <= 300 keypoints/frame, 32-bit windowed matcher) -- a parity case, not the metric.

Only the cpu_baseline leg imports oracle/ (the CPU restatement, timed as a baseline).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "jetracer-orbslam2_amd"))

MODES = {
    "c2": dict(width=640, height=480, cfg=dict(levels=8, cell=8, min_arc=9, max_features=2000),
               match=dict(mode=1, window=-1, max_distance=256),
               workload="640x480 mono, 8-level pyramid, 2000 features/frame (cell 8, FAST-9 t=13, "
                        "top-2000), 256-bit brute-force match t-1->t"),
    "ref": dict(width=640, height=480, cfg=dict(levels=6, cell=32, min_arc=12, max_features=0),
                match=dict(mode=0, window=2, max_distance=4),
                workload="640x480 mono, reference-parity mode (6 levels, cell 32, FAST-12 t=13, "
                         "<=300 keypoints), 32-bit windowed match t-1->t"),
}
FP4_MFMA_PEAK_TFLOPS = 10000.0
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes(width, height, levels, detect_levels, cells, k_out):
    """B_frame = P0 + 2 * sum(P_l) + 52 * K_out, and its per-kernel split."""
    p = [(width >> l) * (height >> l) for l in range(levels)]
    pyramid = p[0] + sum(p)                       # read input, write every level once
    detect = sum(p[:detect_levels]) + 4 * cells   # read detect levels once, write cell keys
    describe = 52 * k_out                         # write records (patch gathers hit L2/MALL)
    frame = p[0] + 2 * sum(p) + 52 * k_out
    return dict(frame=frame, pyramid=pyramid, detect=detect, describe=describe)


def cpu_baseline(frames, mode, seconds=12.0):
    """Oracle (CPU port of the reference semantics) timed on this host: extract + match."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    from concurrent.futures import ThreadPoolExecutor
    m = MODES[mode]
    ocfg = oracle.make_config(m["width"], m["height"], **m["cfg"])
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))  # a 1-GPU box has a 16-CPU share

    def comp(d):
        return ((d == 1).astype(np.uint32) << np.arange(32, dtype=np.uint32)).sum(1).astype(np.uint32)

    def extract(i):
        return oracle.extract_frame(frames[i % len(frames)], ocfg)["records"]

    def match(pair):
        a, b = pair
        pa, pb = np.stack([a["x"], a["y"]], 1), np.stack([b["x"], b["y"]], 1)
        if m["match"]["mode"] == 1:
            oracle.match256(a["desc"], b["desc"], pa, pb, m["match"]["window"], m["match"]["max_distance"])
        else:
            oracle.match_keypoints(pa, comp(a["desc"]), pb, comp(b["desc"]), m["match"]["window"],
                                   m["match"]["max_distance"])

    # time-bounded: chunks of `cores` frames (extract, then match consecutive pairs) until the
    # budget is spent, so the sample stays ~`seconds` whatever the host's real CPU share is
    recs_prev, n_frames, n_pairs, kp = None, 0, 0, 0
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:  # ctypes releases the GIL inside the oracle
        while time.perf_counter() - t0 < seconds:
            recs = list(ex.map(extract, range(n_frames, n_frames + cores)))
            chain = ([recs_prev] if recs_prev is not None else []) + recs
            list(ex.map(match, zip(chain[:-1], chain[1:])))
            n_pairs += len(chain) - 1
            n_frames += len(recs)
            kp += sum(len(r) for r in recs)
            recs_prev = recs[-1]
    dt = time.perf_counter() - t0
    return dict(value=kp / dt, unit="keypoints/s", cores=cores, kind="port",
                sample="%d frames extracted + %d frame pairs matched by the CPU oracle on %d threads "
                       "in %.1f s" % (n_frames, n_pairs, cores, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--mode", choices=sorted(MODES), default="c2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rgb", action="store_true",
                    help="feed interleaved RGB8 frames (SURVEY.md 8f-1): the gray conversion is fused into "
                         "the pyramid kernel; not the BASELINE metric (its configs are grayscale)")
    ap.add_argument("--gather", action="store_true",
                    help="N > 1: also gather every rank's keypoint records to rank 0 each step (asynchronous, "
                         "double-buffered).  Off by default: the frames are independent and the path has no "
                         "exchange step (26.6 MB per rank and step would ride on one xGMI link each)")
    ap.add_argument("--stage-iters", type=int, default=10, help="(unused; kept for old command lines)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import orbfe
    if not os.path.exists(orbfe.LIB_PATH) and int(os.environ.get("LOCAL_RANK", "0")) == 0:
        import __graft_entry__  # clean checkout: compile the HIP library first (hipcc, gfx950)
        __graft_entry__.build()
    from orbfe import synth
    from orbfe.dist import gather_keypoints_async

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # ORBFE_BENCH_SHARE_GPU=1 (rehearsal on a 1-GPU box only): every rank uses cuda:0 and the
        # collectives go over gloo, because RCCL refuses two ranks on one device
        share = os.environ.get("ORBFE_BENCH_SHARE_GPU") == "1"
        if share:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    m = MODES[args.mode]
    w, h, B = m["width"], m["height"], args.batch
    ctx = orbfe.Context(w, h, max_batch=B, device=local_rank, **m["cfg"])
    # synthetic data: 16 distinct corner-rich scenes per rank, repeated to fill the batch
    n_distinct = min(16, B)
    base = synth.frames(w, h, n_distinct, first_index=1000 * rank, kind="rects", **synth.DENSE)
    frames = torch.from_numpy(base).to(dev)[torch.arange(B, device=dev) % n_distinct].contiguous()
    if args.rgb:  # R = G = B = gray scene +- a channel-dependent offset: corners survive the conversion
        off = torch.tensor([3, 0, -3], dtype=torch.int16, device=dev)
        frames = (frames.to(torch.int16).unsqueeze(-1) + off).clamp(0, 255).to(torch.uint8).contiguous()
    # records / counts are double-buffered: the gather of step i (RCCL, asynchronous) overlaps
    # the kernels of step i + 1, which write the other buffer
    recs = [torch.zeros(B * ctx.cap * 52, dtype=torch.uint8, device=dev) for _ in range(2)]
    cnts = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(2)]
    rec, cnt = recs[0], cnts[0]
    idx = torch.zeros(max(B - 1, 1) * ctx.cap, dtype=torch.int32, device=dev)
    dst = torch.zeros(max(B - 1, 1) * ctx.cap, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    mm = m["match"]
    gather_out = None
    do_gather = world > 1 and args.gather
    if do_gather and rank == 0:
        gather_out = [(torch.empty((world, B * ctx.cap * 52), dtype=torch.uint8, device=dev),
                       torch.empty((world, B), dtype=torch.int32, device=dev)) for _ in range(2)]
    pending = [None, None]
    step_no = [0]

    stage_events = []  # per timed step: 5 events bracketing pyramid | detect | describe | match

    def step(record=False):
        b = step_no[0] & 1
        step_no[0] += 1
        if pending[b] is not None:  # the gather that last read this buffer must be done
            pending[b].wait()
            pending[b] = None
        r, c = recs[b], cnts[b]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if record else None
        # orbfe_extract == build_pyramid + detect_batch + describe_batch; issued separately so
        # that HIP events on this stream can time each stage inside the timed region
        if ev: ev[0].record()
        if args.rgb:
            orbfe.check(orbfe.lib().orbfe_build_pyramid_rgb(ctx.handle, frames.data_ptr(), 3 * w, 3 * w * h, B, s),
                        ctx.handle)
        else:
            ctx.build_pyramid(frames.data_ptr(), w, w * h, B, s)
        if ev: ev[1].record()
        ctx.detect_batch(B, s)
        if ev: ev[2].record()
        ctx.describe_batch(B, r.data_ptr(), c.data_ptr(), None, s)
        if ev: ev[3].record()
        ctx.match_batch(r.data_ptr(), c.data_ptr(), B, mm["mode"], mm["window"], mm["max_distance"],
                        idx.data_ptr(), dst.data_ptr(), s)
        if ev:
            ev[4].record()
            stage_events.append(ev)
        if do_gather:
            pending[b] = gather_keypoints_async(r, c, gather_out[b] if gather_out else None, dst=0)

    def drain():
        for b in (0, 1):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    def sync():
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(record=True)
    sync()
    elapsed = time.perf_counter() - t0
    counts = cnt.cpu().numpy().astype(np.int64)
    kp_local = int(counts.sum())
    pairs_local = int((counts[:-1] * counts[1:]).sum())
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([kp_local, pairs_local], dtype=torch.int64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        kp_total, pairs_total = int(tot[0].item()), int(tot[1].item())
    else:
        kp_total, pairs_total = kp_local, pairs_local

    # ---- per-stage device time: HIP events recorded on the kernels' stream inside the timed
    #      region (averaged over the K timed steps)
    out = None
    if rank == 0:
        names = ("pyramid", "detect", "describe", "match")
        ms = {k: 0.0 for k in names}
        for ev in stage_events:
            for i, k in enumerate(names):
                ms[k] += ev[i].elapsed_time(ev[i + 1])
        ms_pyr, ms_det, ms_desc, ms_match = (ms[k] / max(len(stage_events), 1) for k in names)
        detect_levels = sum(1 for l in range(m["cfg"]["levels"]) if (m["cfg"]["cell"] >> l) > 0
                            and (w >> l) > 0 and (h >> l) > 0)
        k_out = kp_local / B
        ab = algorithmic_bytes(w, h, m["cfg"]["levels"], detect_levels, ctx.K, k_out)
        stages = {"pyramid": ms_pyr, "detect": ms_det, "describe": ms_desc, "match": ms_match}
        # matcher bytes, SURVEY.md 8d: 40 B per descriptor+position in, 8 B (idx, dist) out
        c64 = counts.astype(np.int64)
        ab["match"] = float((40 * (c64[:-1] + c64[1:]) + 8 * c64[:-1]).sum()) / B if mm["mode"] == 1 else \
            float((12 * (c64[:-1] + c64[1:]) + 8 * c64[:-1]).sum()) / B
        kernels = {"pyramid": "pyramid_fused_kernel", "detect": "detect_tile_kernel",
                   "describe": "select_kernel+describe_kernel",
                   "match": "match_expand_kernel+match_mfma_kernel" if mm["mode"] == 1 else "match_batch_ref_kernel"}
        traffic_all = {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and B == 256:  # the PMC passes were taken at batch 256
            try:
                traffic_all = json.load(open(tpath)).get(args.mode, {})
            except Exception:
                traffic_all = {}
        per_stage = {k: {"kernel": kernels[k], "ms": stages[k], "algorithmic_bytes": ab[k] * B,
                         "achieved_GBps": ab[k] * B / (stages[k] * 1e-3) / 1e9,
                         "frac": ab[k] * B / (stages[k] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "traffic": traffic_all.get(k)} for k in stages}
        if mm["mode"] == 1:
            # the 256-bit matcher runs on the matrix cores: 2 x 256 flop per pair on e2m1 operands,
            # dense FP4 MFMA peak ~10 PFLOP/s (MI355X_MICROARCH.md, Matrix cores)
            flops = 512.0 * pairs_local
            per_stage["match"]["mfma"] = {"bound": "mfma", "unit": "TFLOP/s", "peak": FP4_MFMA_PEAK_TFLOPS,
                                          "achieved": flops / (ms_match * 1e-3) / 1e12,
                                          "frac": flops / (ms_match * 1e-3) / 1e12 / FP4_MFMA_PEAK_TFLOPS,
                                          "flops_per_launch": flops}
        dom = max(stages, key=lambda k: stages[k])  # the kernel with the largest share of the step
        dom_kernel = kernels[dom]
        achieved = per_stage[dom]["achieved_GBps"]
        traffic = traffic_all.get(dom)
        ms_extract = ms_pyr + ms_det + ms_desc
        out = {
            "metric": "ORB keypoints/sec end-to-end (extract + match), 640x480 8-level",
            "value": kp_total * args.steps / elapsed,
            "unit": "keypoints/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": m["workload"] + (" [RGB8 input, conversion fused]" if args.rgb else ""),
                       "frames_per_gpu_per_step": B,
                       "frames_per_step": B * world, "keypoints_per_frame": k_out,
                       "collective": ("async gather of 52-byte keypoint records to rank 0, overlapped with the "
                                      "next step" if do_gather else
                                      "none in the data path (independent frames); barrier + timing reductions only")},
            "frames_per_s": B * world * args.steps / elapsed,
            "matcher_gpairs_per_s": pairs_local / (ms_match * 1e-3) / 1e9,
            "matcher_pairs_per_step": pairs_total,
            "stage_ms": stages,
            "path_hbm": {"algorithmic_bytes_per_frame": ab["frame"],
                         "achieved_GBps": ab["frame"] * B / (ms_extract * 1e-3) / 1e9,
                         "frac_of_8TBps": ab["frame"] * B / (ms_extract * 1e-3) / 1e9 / HBM_PEAK_GBPS},
            "roofline": {"bound": "hbm", "kernel": dom_kernel, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": ab[dom] * B, "avg_launch_ms": stages[dom],
                         "note": "declared roofline is HBM; the kernels are VALU-issue bound on MI355X "
                                 "(DESIGN.md section 4), so frac stays small by construction",
                         "stages": per_stage},
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is timed at N = 1 only
            out["cpu_baseline"] = cpu_baseline(base, args.mode)
        elif world > 1:
            out["cpu_baseline"] = None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
