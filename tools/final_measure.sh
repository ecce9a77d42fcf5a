#!/bin/bash
# Everything the documents quote, in one gpurun call: tools/final_measure.sh <tag> -> gpurun_out/<tag>/...
TAG=${1:-final}; OUT=gpurun_out/$TAG; mkdir -p $OUT
bash tools/collect_profiles.sh $TAG > $OUT/collect.log 2>&1; echo "collect rc=$?"
for mode in ref c3 c4 c5; do
  timeout -k 10 300 python bench.py --mode $mode --steps 20 --warmup 5 > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err; echo "bench $mode rc=$?"
done
timeout -k 10 300 python bench.py --mode match --steps 20 --warmup 5 > $OUT/bench_match.json 2> $OUT/bench_match.err; echo "bench match rc=$?"
timeout -k 10 300 python bench.py --rotate 3 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_rotate3.json 2> $OUT/bench_rotate3.err; echo "bench rotate rc=$?"
timeout -k 10 300 python bench.py --batch 256 --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $OUT/bench_batch256.json 2> $OUT/bench_batch256.err; echo "bench batch256 rc=$?"
ORBFE_BENCH_FORCE_COMM=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/bench_forcecomm.json 2> $OUT/bench_forcecomm.err; echo "bench forcecomm rc=$?"
MODE=c3 bash tools/collect_profiles.sh ${TAG}_c3 > $OUT/collect_c3.log 2>&1; echo "collect c3 rc=$?"
timeout -k 10 200 python tools/stage_latency.py > $OUT/stage_latency.txt 2>&1; echo "stage latency rc=$?"
timeout -k 10 200 python tools/latency_probe.py > $OUT/latency_probe.txt 2>&1; echo "latency probe rc=$?"
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], "value=%.4g"%d["value"], d["unit"], "ms=%.4f"%d["ms_per_step"], {k:round(v,4) for k,v in d.get("stage_ms",{}).items()})
    except Exception as e: print(f, "ERR", e)
PY
