#!/bin/bash
# align_depth: parity tests of round 4 + the bench under both protocols (quick A/B after a kernel change)
TAG=${1:-r4q}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round2.py -m gpu -x -q -k "align or f3 or f4 or match_port or reproject" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for proto in literal zero; do
  ORBFE_ALIGN_PROTOCOL=$proto timeout -k 10 100 python bench.py --mode align --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_$proto.json 2>> $OUT/bench.err; echo "$proto rc=$?"
done
for c in 32 128 256; do
  ORBFE_ALIGN_CHUNK=$c timeout -k 10 100 python bench.py --mode align --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_chunk$c.json 2>> $OUT/bench.err
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], "ms=%.4f"%d["roofline"]["avg_launch_ms"], "roof %.3f"%d["roofline"]["frac"], "%.4g frames/s"%d["value"])
    except Exception as e: print(f, "ERR", e)
PY
