#!/bin/bash
# round 5: A/B of the register-streamed matcher forms (ORBFE_MATCH_V2=<waves per workgroup><waves per SIMD>) against the product
# usage (through gpurun): tools/r5_match_ab.sh <tag> [variants...]
TAG=${1:-r5match}; shift
VARS=${@:-base 42 41 82 22 21}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $ROOT
for v in $VARS; do
  if [ "$v" = base ]; then unset ORBFE_MATCH_V2; else export ORBFE_MATCH_V2=$v; fi
  if [ "$v" != base ]; then
    timeout -k 10 300 python -m pytest tests -x -q -m gpu -k "match" > $OUT/pytest_$v.log 2>&1; echo "variant $v pytest rc=$?" | tee -a $OUT/summary.txt
    tail -1 $OUT/pytest_$v.log | tee -a $OUT/summary.txt
  fi
  for rep in 1 2; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 40 > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err
    python - <<PY | tee -a $OUT/summary.txt
import json
try:
    d = json.loads(open("$OUT/bench_${v}_$rep.json").read())
    print("variant %-5s rep $rep: step %.4f ms, stages %s" % ("$v", d["ms_per_step"], {k: round(x, 4) for k, x in d["stage_ms"].items()}))
except Exception as e:
    print("variant $v rep $rep: no result (%s)" % e)
PY
  done
done
