#!/bin/bash
# the N > 1 code path of bench.py after round 4's changes: two ranks on one GPU over gloo (c2 and c5), the forced-comm run, and
# the stage-latency tool (now with align_depth_to_other)
OUT=gpurun_out/r4_multi; mkdir -p $OUT
ORBFE_BENCH_SHARE_GPU=1 timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --batch 512 > $OUT/bench_2rank_share.json 2> $OUT/bench_2rank_share.err; echo "share2 c2 rc=$?"
ORBFE_BENCH_SHARE_GPU=1 timeout -k 10 300 python bench.py --gpus 2 --mode c5 --steps 5 --warmup 2 --no-extras > $OUT/bench_c5_2rank_share.json 2> $OUT/bench_c5_2rank_share.err; echo "share2 c5 rc=$?"
ORBFE_BENCH_SHARE_GPU=1 timeout -k 10 300 python bench.py --gpus 2 --mode c4 --steps 5 --warmup 2 --no-extras --exact-gather > $OUT/bench_c4_2rank_share.json 2> $OUT/bench_c4_2rank_share.err; echo "share2 c4 exact rc=$?"
timeout -k 10 200 python tools/stage_latency.py > $OUT/stage_latency.txt 2>&1; echo "stage latency rc=$?"; grep -v amdgpu $OUT/stage_latency.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], "n_gpus", d["n_gpus"], "value=%.4g"%d["value"], "ms=%.4f"%d["ms_per_step"], d["config"]["collective"][:60], d.get("gather",{}).get("link_utilisation_one_way"))
    except Exception as e: print(f, "ERR", e)
PY
