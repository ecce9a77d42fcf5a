"""bench.py's launcher logic on CPU: `--gpus N` must start N ranks (it ignored N in round 1)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    if env:
        e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=e, capture_output=True, text=True,
                          timeout=120)


def test_gpus_n_dry_run_builds_n_child_ranks():
    r = _run("--gpus", "2", "--steps", "5", "--dry-run")
    assert r.returncode == 0, r.stderr
    kids = json.loads(r.stdout)
    assert len(kids) == 2
    for rank, k in enumerate(kids):
        assert k["env"]["RANK"] == str(rank) and k["env"]["LOCAL_RANK"] == str(rank)
        assert k["env"]["WORLD_SIZE"] == "2" and k["env"]["MASTER_ADDR"] == "127.0.0.1"
        assert k["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        assert k["argv"][1].endswith("bench.py") and "--dry-run" not in k["argv"]
        assert k["argv"][k["argv"].index("--gpus") + 1] == "2"
    assert kids[0]["env"]["MASTER_PORT"] == kids[1]["env"]["MASTER_PORT"]


def test_gpus_must_match_world_size():
    # under a launcher (WORLD_SIZE set) a mismatching --gpus is an error, not a silent 1-GPU run
    r = _run("--gpus", "4", "--steps", "1", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "--gpus 4 but WORLD_SIZE is 2" in r.stderr


def test_dist_header_and_binding_agree():
    import re
    from orbfe import dist as od
    text = open(os.path.join(ROOT, "include", "orbfe_dist.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(orbfe_dist_[a-z0-9_]+)\s*\(", text)))
    assert declared == sorted(od.DIST_EXPORTS)
    lib = od.dist_lib()
    for name in declared:
        assert hasattr(lib, name), name
    import ctypes as C
    b, e = C.c_int(), C.c_int()
    for n in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            for rank in range(world):
                assert lib.orbfe_dist_shard_range(n, rank, world, C.byref(b), C.byref(e)) == 0
                assert (b.value, e.value) == od.shard_range(n, rank, world)
    assert lib.orbfe_dist_shard_range(10, 2, 2, C.byref(b), C.byref(e)) != 0
