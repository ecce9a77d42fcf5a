#!/bin/bash
# One gpurun call: GPU tests, then the bench modes.  Logs under gpurun_out/<tag>/.
TAG=${1:-check}
OUT=gpurun_out/$TAG
mkdir -p $OUT
set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/pytest.log
for mode in c2 ref c3 c4 c5; do
  timeout -k 10 300 python bench.py --mode $mode --steps 20 --warmup 3 > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err; echo "bench $mode rc=$?" | tee -a $OUT/summary.txt
done
ORBFE_BENCH_SHARE_GPU=1 timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-extras > $OUT/bench_2rank_share.json 2> $OUT/bench_2rank_share.err; echo "bench share2 rc=$?" | tee -a $OUT/summary.txt
ORBFE_BENCH_SHARE_GPU=1 timeout -k 10 300 python bench.py --gpus 2 --mode c5 --steps 5 --warmup 2 --no-extras > $OUT/bench_c5_2rank_share.json 2> $OUT/bench_c5_2rank_share.err; echo "bench c5 share2 rc=$?" | tee -a $OUT/summary.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], "value=%.4g"%d["value"], "ms=%.4f"%d["ms_per_step"], "n_gpus",d["n_gpus"], {k:round(v,4) for k,v in d["stage_ms"].items()})
    except Exception as e: print(f, "ERR", e)
PY
