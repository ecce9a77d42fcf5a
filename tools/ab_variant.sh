#!/bin/bash
# A/B of a kernel variant (tools/build_variant.sh <name> ...) against the product library on one box:
#   tools/ab_variant.sh <name> [pytest -k expression]   -> parity suite on the variant, then 2 x (base, variant) bench steps
V=$1; K=${2:-"extract or detect or fuzz or ext_regime or stereo or shard"}; R=$(pwd); OUT=gpurun_out/ab_$V; mkdir -p $OUT
ORBFE_LIB=$R/jetracer-orbslam2_amd/.variants/$V/liborbfe.so timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "$K" > $OUT/pytest.log 2>&1; echo "pytest($V) rc=$?"; tail -2 $OUT/pytest.log
for v in base $V base $V; do
  if [ $v = base ]; then unset ORBFE_LIB; else export ORBFE_LIB=$R/jetracer-orbslam2_amd/.variants/$v/liborbfe.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'ms/step %.4f' % d['ms_per_step'], {k: round(x,4) for k,x in d['stage_ms'].items()}, 'kp/frame %.1f' % d['config']['keypoints_per_frame'])" | tee -a $OUT/ab.txt
done
