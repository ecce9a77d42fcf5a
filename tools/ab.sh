#!/bin/bash
# A/B timing of liborbfe.so variants on one box: tools/ab.sh <tag> <variant...>  ("base" = the in-tree library)
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then unset ORBFE_LIB; else export ORBFE_LIB=$PWD/jetracer-orbslam2_amd/.variants/$v/liborbfe.so; fi
  timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $OUT/$v.$rep.json 2>$OUT/$v.$rep.err
  python -c "
import json; d=json.load(open('$OUT/$v.$rep.json')); print('%-12s rep$rep ms=%.4f'%('$v',d['ms_per_step']), {k:round(x,4) for k,x in d['stage_ms'].items()})"
done; done
