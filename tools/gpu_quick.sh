#!/bin/bash
# quick GPU check: parity tests + c2 bench (optionally an env A/B).  usage: gpu_quick.sh <tag> [pytest -k expr]
TAG=${1:-quick}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q ${2:+-k "$2"} > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$OUT/bench.json')); print('value=%.4g ms=%.4f'%(d['value'],d['ms_per_step']), {k:round(v,4) for k,v in d['stage_ms'].items()})"
