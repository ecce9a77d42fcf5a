#!/bin/bash
# round 4, one gpurun call: GPU tests, the align bench (+ its kernel trace), the c2 bench with the fixed-mode extras
TAG=${1:-r4a}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout -k 10 200 python bench.py --mode align --steps 10 --warmup 3 > $OUT/bench_align.json 2> $OUT/bench_align.err; echo "align rc=$?"
for c in 8 16 32 128 1024; do
  ORBFE_ALIGN_CHUNK=$c timeout -k 10 100 python bench.py --mode align --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_align_chunk$c.json 2>> $OUT/bench_align.err; echo "align chunk $c rc=$?"
done
ORBFE_ALIGN_PROTOCOL=literal timeout -k 10 100 python bench.py --mode align --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_align_literal.json 2>> $OUT/bench_align.err; echo "align literal rc=$?"
R=$(pwd); (cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/trace_align -o run -- python3 $R/bench.py --mode align --steps 10 --warmup 3 --no-cpu-baseline > $R/$OUT/trace_align.log 2>&1); echo "trace rc=$?"
cp "$(find $OUT/trace_align -name '*kernel_stats.csv' | head -1)" $OUT/align_kernel_stats.csv 2>/dev/null; rm -rf $OUT/trace_align
timeout -k 10 400 python bench.py --steps 20 --warmup 3 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "c2 rc=$?"
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], "value=%.4g"%d["value"], d["unit"], "ms=%.4f"%d["ms_per_step"], "roof %.3f"%d["roofline"]["frac"], {k:round(v,4) for k,v in d.get("stage_ms",{}).items()})
        if "fixed_modes" in d:
            for k,v in d["fixed_modes"].items():
                if isinstance(v,dict): print("   fixed", k, "%.4g"%v["value"], "ms=%.4f"%v["ms_per_step"], {a:round(b,4) for a,b in v["stage_ms"].items()}, v["kernels"])
    except Exception as e: print(f, "ERR", e)
PY
cat $OUT/align_kernel_stats.csv 2>/dev/null | head -8
