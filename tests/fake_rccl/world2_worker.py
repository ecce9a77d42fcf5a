"""TEST INFRASTRUCTURE: one rank of a world > 1 run of liborbfe_dist.so on ONE GPU, over the loopback transport of
fake_rccl.cpp (the parent test starts `world` of these with tests/fake_rccl/_build in front of LD_LIBRARY_PATH, so that
the shipped liborbfe_dist.so resolves librccl.so.1 there).  No torch in this process: torch brings its own librccl.

    python world2_worker.py <rank> <world> <scratch dir>

Every rank extracts its shard with the HIP library (orbfe_extract on device 0) and drives the entry points of
include/orbfe_dist.h whose bodies sit under `world > 1`; what arrives is written to <scratch dir> for the parent, which
compares it with the CPU oracle:
  fixed.npz   orbfe_dist_gather_keypoints, fixed stride, root 0, the root extracting straight into its own block
  exact.npz   the exact-length form (+ orbfe_dist_exact_offsets), root 0
  root1.npz   fixed stride with root = world - 1 (root placement / per-rank offsets with a non-zero root)
  c5_<r>.npz  tile-sharded detection of one frame: export -> orbfe_dist_allreduce_max_keys -> import -> describe, per rank
  host_<r>.json  orbfe_dist_host_allreduce (max, sum) + orbfe_dist_barrier
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "jetracer-orbslam2_amd"))

W, H, N_TOTAL = 320, 240, 6
CFG = dict(levels=4, cell=8, min_arc=9, max_features=300)
C5 = dict(width=640, height=480, cfg=dict(levels=8, cell=16, min_arc=9, max_features=700))


def scene(i):
    from orbfe import synth
    dense = i % 2 == 0  # ragged counts: dense scenes fill the budget, sparse ones do not
    return synth.frame(W, H, 900 + i, "rects", n_rects=200 if dense else 12, min_size=6, max_size=24 if dense else None)


def c5_scene():
    from orbfe import synth
    return synth.frame(C5["width"], C5["height"], 950, "rects", **synth.DENSE)


class Hip:
    def __init__(self):
        self.lib = C.CDLL("libamdhip64.so")
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipFree.argtypes = [C.c_void_p]
        self.lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.lib.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]

    def ok(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: hipError %d" % (what, rc))

    def malloc(self, n, fill=None):
        p = C.c_void_p()
        self.ok(self.lib.hipMalloc(C.byref(p), max(n, 1)), "hipMalloc")
        if fill is not None:
            self.ok(self.lib.hipMemset(p, fill, max(n, 1)), "hipMemset")
        return p.value

    def up(self, arr):
        a = np.ascontiguousarray(arr)
        p = self.malloc(a.nbytes)
        self.ok(self.lib.hipMemcpy(p, a.ctypes.data, a.nbytes, 1), "hipMemcpy H2D")
        return p

    def down(self, p, dtype, count):
        out = np.empty(count, dtype=dtype)
        self.ok(self.lib.hipMemcpy(out.ctypes.data, p, out.nbytes, 2), "hipMemcpy D2H")
        return out

    def sync(self):
        self.ok(self.lib.hipDeviceSynchronize(), "hipDeviceSynchronize")


def main():
    rank, world, scratch = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    assert "torch" not in sys.modules
    import orbfe
    from orbfe.dist import RcclComm, shard_range
    assert "torch" not in sys.modules, "orbfe / orbfe.dist must not pull torch into this process"
    hip = Hip()

    def exchange(ident):  # rank 0's unique id reaches the others through a file
        path = os.path.join(scratch, "id.bin")
        if rank == 0:
            with open(path + ".tmp", "wb") as f:
                f.write(ident)
            os.rename(path + ".tmp", path)
            return ident
        t0 = time.time()
        while not os.path.exists(path):
            time.sleep(0.01)
            assert time.time() - t0 < 60, "rank 0 never wrote the unique id"
        return open(path, "rb").read()

    comm = RcclComm(rank, world, 0, exchange)  # every rank on HIP device 0
    with open("/proc/self/maps") as f:
        loaded = sorted({line.split()[-1] for line in f if "librccl" in line})
    assert loaded and all("fake_rccl/_build" in p for p in loaded), "the loopback transport is not what got loaded: %r" % loaded

    # ---------------------------------------------------------------- gathers
    f0, f1 = shard_range(N_TOTAL, rank, world)
    n = f1 - f0
    ctx = orbfe.Context(W, H, max_batch=n, device=0, **CFG)
    cap = ctx.cap
    frames = np.stack([scene(i) for i in range(f0, f1)])
    d_in = hip.up(frames)
    d_cnt = hip.malloc(4 * n, 0)
    d_rec = hip.malloc(52 * n * cap, 0)
    for name, root, exact in (("fixed", 0, 0), ("exact", 0, 1), ("root1", world - 1, 0)):
        is_root = rank == root
        d_all_rec = hip.malloc(52 * N_TOTAL * cap, 0xEE) if is_root else None
        d_all_cnt = hip.malloc(4 * N_TOTAL, 0xEE) if is_root else None
        in_place = is_root and name == "fixed"  # the root extracts straight into its own block: no local copy
        r_ptr = d_all_rec + 52 * root * n * cap if in_place else d_rec
        c_ptr = d_all_cnt + 4 * root * n if in_place else d_cnt
        ctx.extract(d_in, W, W * H, n, r_ptr, c_ptr, None, 0)
        comm.gather_keypoints(r_ptr, c_ptr, n, cap, d_all_rec, d_all_cnt, root, exact, 0)
        t = comm.ticket()
        comm.wait_ticket(t, 0)
        out = {}
        if is_root and exact:
            d_off = hip.malloc(8 * N_TOTAL, 0xEE)
            comm.exact_offsets(d_all_cnt, n, cap, d_off, 0)
            hip.sync()
            out["offsets"] = hip.down(d_off, np.int64, N_TOTAL)
            hip.lib.hipFree(d_off)
        comm.sync()
        hip.sync()
        if is_root:
            out["records"] = hip.down(d_all_rec, np.uint8, 52 * N_TOTAL * cap)
            out["counts"] = hip.down(d_all_cnt, np.int32, N_TOTAL)
            np.savez(os.path.join(scratch, name + ".npz"), cap=cap, **out)
            hip.lib.hipFree(d_all_rec)
            hip.lib.hipFree(d_all_cnt)
    ctx.close()

    # ---------------------------------------------------------------- C5 flow: tile-sharded detection of ONE frame
    w5, h5 = C5["width"], C5["height"]
    c5 = orbfe.Context(w5, h5, max_batch=1, device=0, **C5["cfg"])
    d_f = hip.up(c5_scene())
    d_keys = hip.malloc(4 * c5.K, 0)
    d_r5 = hip.malloc(52 * c5.cap, 0)
    d_c5 = hip.malloc(4, 0)
    c5.build_pyramid(d_f, w5, w5 * h5, 1, 0)
    c5.detect_batch_shard(1, rank, world, 0)
    c5.export_cell_keys(1, d_keys, 0)
    partial = hip.down(d_keys, np.uint32, c5.K)
    comm.allreduce_max_keys(d_keys, c5.K, 0)
    comm.wait(0)
    c5.import_cell_keys(1, d_keys, 0)
    c5.describe_batch(1, d_r5, d_c5, None, 0)
    hip.sync()
    np.savez(os.path.join(scratch, "c5_%d.npz" % rank), cap=c5.cap, partial_keys=partial,
             merged_keys=hip.down(d_keys, np.uint32, c5.K), records=hip.down(d_r5, np.uint8, 52 * c5.cap),
             count=hip.down(d_c5, np.int32, 1))
    c5.close()

    # ---------------------------------------------------------------- host reductions
    mx = comm.host_allreduce([rank + 1.0, 10.0 * rank, -3.5], "max")
    sm = comm.host_allreduce([rank + 1.0, 10.0 * rank, -3.5], "sum")
    comm.barrier()
    with open(os.path.join(scratch, "host_%d.json" % rank), "w") as f:
        json.dump({"max": mx, "sum": sm, "librccl": loaded}, f)
    comm.close()
    print("rank %d of %d done" % (rank, world))


if __name__ == "__main__":
    main()
