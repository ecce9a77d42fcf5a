#!/bin/bash
# Build the profiling variants the evidence scripts use, from tools/experiments/profiling_probes.patch applied to a scratch
# copy of the sources (the product sources hold none of these macros; every one of these builds gives WRONG results).
# usage: tools/build_probe_variants.sh [detect] [describe] [align] [ring]      (default: all)   -- run here, before gpurun
set -e
cd "$(dirname "$0")/.."
WHAT=${@:-detect describe align ring}
for w in $WHAT; do
  case $w in
    detect)   for n in 1 2 3; do tools/build_variant.sh det$n -p profiling_probes -DORBFE_DETECT_STOP_AFTER=$n; done ;;
    describe) for n in 1 2 3; do tools/build_variant.sh desc$n -p profiling_probes -DORBFE_DESCRIBE_STOP_AFTER=$n; done
              tools/build_variant.sh descnostage -p profiling_probes -DORBFE_DESCRIBE_NOSTAGE ;;
    align)    for n in 1 2 3; do tools/build_variant.sh al$n -p profiling_probes -DORBFE_ALIGN_ABLATE=$n; done ;;
    ring)     tools/build_variant.sh ringnolds -p profiling_probes -DORBFE_DETECT_RING_NOLDS
              tools/build_variant.sh rows3 -p profiling_probes -DORBFE_DETECT_RING_ROWS3 ;;
  esac
done
